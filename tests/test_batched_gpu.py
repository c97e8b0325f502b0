"""GPU: several images per launch (gi2d_train_steps_batched, gi2d_fast_rasterize_forward_backward_batched;
csrc/gi2d_batch.h) -- every image's results must be those of its own single-image calls, BIT FOR BIT: the batched
kernels run the single-image kernels' code on an argument block picked by workgroup index."""
import ctypes as C
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"

STATE_ROWS = ("xyz", "chol", "feat", "m_xyz", "v_xyz", "m_chol", "v_chol", "m_feat", "v_feat")


def _fitters(kind, optimizer, sizes, track_best=True, **kw):
    from gaussianimage_plus_amd.launch import synthetic_image
    from gaussianimage_plus_amd.trainer import NativeFitter
    out = []
    for i, (h, w, n) in enumerate(sizes):
        gt = synthetic_image(h, w, 60 + i).to(DEV)
        out.append(NativeFitter(gt, n, kind=kind, lr=0.01, seed=11 + i, optimizer=optimizer, track_best=track_best,
                                eps=1e-15 if optimizer == "adan" else 1e-8, **kw))
    return out


def _assert_same(a, b, tag):
    assert a.n == b.n and a.iteration == b.iteration, tag
    for nm in STATE_ROWS + (("_d_xyz", "_pg_chol") if a.optimizer == "adan" else ()):
        ta, tb = getattr(a, nm)[:a.n], getattr(b, nm)[:b.n]
        assert torch.equal(ta, tb), f"{tag}: {nm} differs in {int((ta != tb).sum())} elements"
    assert torch.equal(a.out_img, b.out_img), f"{tag}: render"
    assert torch.equal(a.tile_sse, b.tile_sse), f"{tag}: per-tile squared errors"
    assert torch.equal(a.xys[:a.n], b.xys[:b.n]) and torch.equal(a.radii[:a.n], b.radii[:b.n]), f"{tag}: projection"
    if a.track_best:
        assert torch.equal(a.best_sse, b.best_sse) and torch.equal(a.best_info, b.best_info), f"{tag}: best snapshot"
        nb = int(a.best_info[0])
        for nm in ("best_xyz", "best_chol", "best_feat"):
            assert torch.equal(getattr(a, nm)[:nb], getattr(b, nm)[:nb]), f"{tag}: {nm}"


# same tile count for every image (portrait + landscape: the batched kernel divides), and ragged mixes (table lookup)
UNIFORM = [(96, 160, 1500), (160, 96, 2100), (96, 160, 700)]
MIXED = [(96, 160, 1500), (50, 70, 400), (128, 128, 2500), (33, 200, 900)]


# each image alone runs its per-gaussian kernels in 64-lane workgroups (N <= 32768), the batch of three in 256-lane ones
BIG = [(256, 384, 15000), (384, 256, 14000), (256, 384, 13000)]
# more than 2048 tiles per image: the best-model decision sums the per-tile errors in four chunks, one after the other in a
# 64-lane workgroup (each image alone), side by side in a 256-lane one (the batch)
LARGE = [(800, 1104, 15000), (1104, 800, 14000), (800, 1104, 13000)]


@pytest.mark.parametrize("kind,optimizer,sizes", [("cholesky", "adan", UNIFORM), ("covariance", "adam", MIXED),
                                                  ("scale_rot", "adam", MIXED), ("cholesky", "adam", MIXED[:1]),
                                                  ("cholesky", "adam", BIG), ("covariance", "adam", BIG),
                                                  ("cholesky", "adam", LARGE)])
def test_batched_iterations_equal_single_image_calls(kind, optimizer, sizes):
    """9 iterations as 1 + 3 + 5 (stretches: the update kernel also starts the next iteration) of K images in one
    launch per kernel == the same iterations of every image alone."""
    from gaussianimage_plus_amd.trainer import BatchFitter
    alone, together = _fitters(kind, optimizer, sizes), _fitters(kind, optimizer, sizes)
    batch = BatchFitter(together)
    for count in (1, 3, 5):
        for f in alone:
            f.train(count)
        batch.train(count)
    torch.cuda.synchronize()
    for i, (a, b) in enumerate(zip(alone, together)):
        a.check_status(), b.check_status()
        _assert_same(a, b, f"{kind}/{optimizer} image {i}")
        assert int(a.best_info[1]) > 0  # a snapshot was taken: the decision logic ran


def test_a_fit_that_changes_between_single_image_and_batched_calls_equals_the_fit_alone():
    """The update kernel of ONE image lets a gaussian enter a tile through the tile's inbox (csrc/gi2d_fast_internal.h::
    Inbox), and only the tile pass issued right behind it, in the same call, is built to take entrants out of one; the
    kernels of a batch append through the row headers.  The two must leave the same rows -- 2 iterations alone, 3 in a
    batch, 4 alone == 9 alone, bit for bit (lr 0.01: gaussians enter tiles on every step) -- and every call must return
    with all inboxes empty (its last update kernel does not bin: whoever comes next finds nothing waiting)."""
    from gaussianimage_plus_amd.trainer import BatchFitter
    sizes = [(256, 384, 15000), (384, 256, 14000)]
    alone, mixed = _fitters("cholesky", "adam", sizes), _fitters("cholesky", "adam", sizes)
    for f in alone:
        f.train(9)
    for f in mixed:
        f.train(2)
        assert _inbox_bits(f) == 0
    BatchFitter(mixed).train(3)
    for f in mixed:
        assert _inbox_bits(f) == 0
        f.train(4)
        assert _inbox_bits(f) == 0
    torch.cuda.synchronize()
    for i, (a, b) in enumerate(zip(alone, mixed)):
        a.check_status(), b.check_status()
        _assert_same(a, b, f"alone / batched / alone, image {i}")


def _wild_fitter(**kw):
    """One 384x256 image whose gaussians use every way into a tile: 4 000 small ones that drift into neighbouring tiles
    (the inbox), 600 piled into one tile (ranks beyond the 256-entry cap: no rank to go by), 400 large ones on ~25 tiles
    (boxes of more than eight tiles keep the row header's atomic), and a learning rate that makes some of them jump more
    than one tile per step (no neighbouring old tile)."""
    from gaussianimage_plus_amd.launch import synthetic_image
    from gaussianimage_plus_amd.trainer import NativeFitter
    h, w, n = 256, 384, 5000
    rng = np.random.default_rng(321)
    u = rng.uniform(-0.98, 0.98, (n, 2))
    u[4000:4600, 0] = (rng.uniform(40, 48, 600) + 0.5) / (0.5 * w) - 1.0
    u[4000:4600, 1] = (rng.uniform(40, 48, 600) + 0.5) / (0.5 * h) - 1.0
    L = np.stack([rng.uniform(0.8, 1.6, n), rng.uniform(-0.3, 0.3, n), rng.uniform(0.8, 1.6, n)], 1)
    L[4600:, 0] = L[4600:, 2] = rng.uniform(10, 14, 400)
    init = {"xyz": torch.from_numpy(np.arctanh(u).astype(np.float32)), "chol": torch.from_numpy(L.astype(np.float32)),
            "feat": torch.from_numpy(rng.uniform(0, 0.2, (n, 3)).astype(np.float32)), "bound": torch.tensor([0.5, 0.0, 0.5])}
    return NativeFitter(synthetic_image(h, w, 75).to(DEV), n, kind="cholesky", lr=0.12, seed=5, init=init, **kw)


def test_every_way_into_a_tile_gives_the_rows_of_the_plain_appends():
    """A single-image call delivers entering gaussians through the tiles' inboxes where it can and through the row
    headers where it cannot (csrc/gi2d_fast_internal.h::Inbox: no neighbouring old tile, no rank, an old box of more than
    eight tiles), in one and the same update kernel; a batch of ONE image runs the kernels that only know the headers.
    Same bits after 1 + 2 + 3 + 4 iterations of a scene that takes every way (_wild_fitter; tools/inbox_stats.py on this
    scene: of ~3 900 entered tiles per step 3 200 go through an inbox -- 750 of them as a second or later entrant of one
    lane's eight slots -- and 700 through the header, for each of the three reasons)."""
    from gaussianimage_plus_amd.trainer import BatchFitter
    alone, one = _wild_fitter(), _wild_fitter()
    batch = BatchFitter([one])
    for count in (1, 2, 3, 4):
        alone.train(count)
        batch.train(count)
        assert _inbox_bits(alone) == 0
    torch.cuda.synchronize()
    alone.check_status(), one.check_status()
    _assert_same(alone, one, "inboxes and headers / headers only")


def _inbox_bits(fit) -> int:
    """Entrants waiting in the inboxes of a fitter's workspace: the set bits of the 64 bitmap words behind each tile
    row's 16 header words and 1024 ids (csrc/gi2d_fast_internal.h: GI2D_FAST_HDR, GI2D_FAST_C, GI2D_INBOX_WORDS)."""
    gp, bp = C.c_void_p(), C.c_void_p()
    from gaussianimage_plus_amd import _lib
    _lib.call("gi2d_fast_workspace_views", fit.ws.data_ptr(), fit.ws.numel(), fit.cap, fit.tx, fit.ty, C.byref(gp),
              C.byref(bp))
    torch.cuda.synchronize()
    off, tiles, lrow = gp.value - fit.ws.data_ptr(), fit.tx * fit.ty, 16 + 1024 + 64
    rows = fit.ws[off:off + 4 * tiles * lrow].view(torch.int32).view(tiles, lrow)
    words = rows[:, 16 + 1024:].cpu().numpy().view(np.uint32)
    return int(np.unpackbits(words.view(np.uint8)).sum())


@pytest.mark.parametrize("k", [8, 11])
def test_xcd_mapped_batch_equals_single_image_calls(k):
    """Eight or more images with the same tile count: image i's tiles go to the workgroups b with b % 8 == i % 8 (one
    image per XCD's L2); with 11 images three XCDs carry a second image and five run one idle slot.  Same bits."""
    from gaussianimage_plus_amd.trainer import BatchFitter
    sizes = [((96, 160) if i % 3 else (160, 96)) + (500 + 37 * i,) for i in range(k)]  # landscape and portrait: 60 tiles
    alone, together = _fitters("covariance", "adam", sizes), _fitters("covariance", "adam", sizes)
    batch = BatchFitter(together)
    for count in (2, 4):
        for f in alone:
            f.train(count)
        batch.train(count)
    torch.cuda.synchronize()
    for i, (a, b) in enumerate(zip(alone, together)):
        a.check_status(), b.check_status()
        _assert_same(a, b, f"xcd-mapped image {i} of {k}")


def test_batched_adaptive_schedule_equals_single_image_schedules():
    """The per-image loop of train.py:120-160 (prune every 10, grow every 20, population on the device) for three images
    in lockstep == the three loops alone: populations, parameters, snapshots."""
    from gaussianimage_plus_amd.trainer import BatchFitter
    sizes = [(96, 144, 400), (144, 96, 300), (64, 80, 350)]
    kw = dict(max_points=3000, device_resident=True)
    alone, together = _fitters("covariance", "adam", sizes, **kw), _fitters("covariance", "adam", sizes, **kw)
    for f in alone:
        f.fit(90, prune_iter=10, grow_iter=20)
    BatchFitter(together).fit(90, prune_iter=10, grow_iter=20)
    torch.cuda.synchronize()
    for i, (a, b) in enumerate(zip(alone, together)):
        a.check_status(), b.check_status()
        assert a.n > sizes[i][2], "the schedule must have grown the population"
        _assert_same(a, b, f"adaptive image {i}")
        assert torch.equal(a.dens_counts, b.dens_counts)


def test_batched_launcher_equals_streams_launcher():
    """launch.fit_images_native with batched=True (one launch per kernel for all images) gives every image the result of
    the K-streams form."""
    from gaussianimage_plus_amd.launch import fit_images_native, synthetic_image
    gts = [synthetic_image(96, 144, 40 + i).to(DEV) for i in range(3)]
    kw = dict(lr=0.018, kind="covariance", max_points=1800, prune_iter=50, grow_iter=100, eps=1e-15, eval_renders=1)
    a = fit_images_native(gts, 1200, 350, batched=False, **kw)
    b = fit_images_native(gts, 1200, 350, batched=True, **kw)
    for ra, rb in zip(a, b):
        assert ra["mse"] == rb["mse"] and ra["num_gaussians"] == rb["num_gaussians"]
        assert ra["psnr"] > 20


@pytest.mark.parametrize("groups", [2, 3, 7])
def test_batches_on_streams_equal_one_batch(groups):
    """batched=G: the images of a GPU as G batches (image i in batch i mod G), each on its own HIP stream and host thread --
    what the launcher and bench.py's images/s leg use; every image's result is that of the single batch (hence of fitting
    it alone).  Five images: uneven batches, and more batches asked for than there are images."""
    from gaussianimage_plus_amd.launch import fit_images_native, synthetic_image
    gts = [synthetic_image(96, 144, 60 + i).to(DEV) for i in range(5)]
    kw = dict(lr=0.018, kind="covariance", max_points=1800, prune_iter=50, grow_iter=100, eps=1e-15, eval_renders=1)
    a = fit_images_native(gts, 1200, 350, batched=True, **kw)
    b = fit_images_native(gts, 1200, 350, batched=groups, **kw)
    for ra, rb in zip(a, b):
        assert ra["mse"] == rb["mse"] and ra["num_gaussians"] == rb["num_gaussians"]
        assert ra["final_num_gaussians"] == rb["final_num_gaussians"]


class _FastImage(C.Structure):
    """struct gi2d_fast_image (include/gi2d.h), field for field."""
    _fields_ = [("num_points", C.c_int), ("tiles_x", C.c_int), ("tiles_y", C.c_int), ("img_width", C.c_uint),
                ("img_height", C.c_uint), ("grad_scale", C.c_float), ("v_output", C.c_void_p), ("target", C.c_void_p),
                ("tile_sse", C.c_void_p), ("workspace", C.c_void_p), ("workspace_bytes", C.c_size_t),
                ("status", C.c_void_p), ("out_img", C.c_void_p)]


@pytest.mark.parametrize("given", [True, False])
def test_batched_tile_pass_equals_single_image_passes(given):
    """gi2d_fast_rasterize_forward_backward_batched on three scenes == three gi2d_fast_rasterize_forward_backward calls:
    images, per-tile errors and (after the reduce) all four gradients, with the gradient image given and with the L2
    gradient formed from a target."""
    import sys, os
    sys.path.insert(0, os.path.dirname(__file__))
    from helpers import synth_cholesky, synth_gt
    from oracle import oracle as O
    import gaussianimage_plus_amd.gsplat.cuda as _C
    from gaussianimage_plus_amd import _lib
    lib = _lib.load()
    st = torch.cuda.current_stream().cuda_stream
    scenes = []
    for i, (h, w, n) in enumerate([(96, 160, 3000), (70, 50, 500), (128, 144, 6000)]):
        xyz, L, col, op = synth_cholesky(n, h, w, 3 + i)
        tb = O.tile_bounds(h, w)
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
        xys, depths, radii, conics, nth = _C.project_gaussians_2d_forward(n, 3.0, t(xyz), t(L), h, w, tb, 0.01, 1.0, False)
        rng = np.random.default_rng(i)
        aux = t(rng.normal(size=(h, w, 3)).astype(np.float32) / (h * w)) if given else t(synth_gt(h, w, 9 + i))
        scenes.append(dict(h=h, w=w, n=n, tb=tb, xys=xys, radii=radii, conics=conics, col=t(col), op=t(op), aux=aux))

    def run(batched):
        res, imgs = [], (_FastImage * len(scenes))()
        for k, sc in enumerate(scenes):
            ws = _C.FastWorkspace(sc["n"], sc["tb"], sc["xys"])
            out = torch.empty(sc["h"], sc["w"], 3, device=DEV)
            sse = torch.zeros(sc["tb"][0] * sc["tb"][1], device=DEV)
            _lib.call("gi2d_fast_bin", sc["n"], sc["xys"].data_ptr(), sc["radii"].data_ptr(), sc["conics"].data_ptr(),
                      sc["col"].data_ptr(), sc["op"].data_ptr(), sc["tb"][0], sc["tb"][1], 1.0, ws.buf.data_ptr(),
                      ws.buf.numel(), ws.status.data_ptr(), st)
            gs = 2.0 / (3.0 * sc["h"] * sc["w"])
            if batched:
                imgs[k] = _FastImage(sc["n"], sc["tb"][0], sc["tb"][1], sc["w"], sc["h"], 0.0 if given else gs,
                                     sc["aux"].data_ptr() if given else None, None if given else sc["aux"].data_ptr(),
                                     None if given else sse.data_ptr(), ws.buf.data_ptr(), ws.buf.numel(),
                                     ws.status.data_ptr(), out.data_ptr())
            else:
                _lib.call("gi2d_fast_rasterize_forward_backward", sc["n"], sc["tb"][0], sc["tb"][1], sc["w"], sc["h"],
                          None, sc["aux"].data_ptr() if given else None, None if given else sc["aux"].data_ptr(),
                          0.0 if given else gs, None if given else sse.data_ptr(), ws.buf.data_ptr(), ws.buf.numel(),
                          ws.status.data_ptr(), out.data_ptr(), st)
            res.append(dict(ws=ws, out=out, sse=sse))
        if batched:
            table = torch.empty(int(lib.gi2d_batch_bytes(len(scenes))), dtype=torch.uint8, device=DEV)
            _lib.call("gi2d_fast_rasterize_forward_backward_batched", len(scenes), imgs, table.data_ptr(), table.numel(),
                      st)
        for sc, r in zip(scenes, res):
            g = [torch.empty(sc["n"], c, device=DEV) for c in (2, 3, 3, 1)]
            _lib.call("gi2d_fast_rasterize_backward_reduce", sc["n"], sc["tb"][0], sc["tb"][1], r["ws"].buf.data_ptr(),
                      r["ws"].buf.numel(), g[0].data_ptr(), g[1].data_ptr(), g[2].data_ptr(), g[3].data_ptr(), None, st)
            r["grads"] = g
        torch.cuda.synchronize()
        return res

    for k, (a, b) in enumerate(zip(run(False), run(True))):
        assert a["ws"].status.tolist()[:2] == b["ws"].status.tolist()[:2] == [1, 0]
        assert torch.equal(a["out"], b["out"]), f"scene {k}: image"
        assert torch.equal(a["sse"], b["sse"]), f"scene {k}: tile errors"
        for ga, gb in zip(a["grads"], b["grads"]):
            assert torch.equal(ga, gb) and bool(torch.isfinite(ga).all()), f"scene {k}: gradients"
        assert float(a["grads"][2].abs().sum()) > 0


def test_batched_entry_rejects_bad_batches():
    from gaussianimage_plus_amd import _lib
    from gaussianimage_plus_amd.trainer import BatchFitter
    a = _fitters("cholesky", "adam", [(64, 64, 200)])[0]
    b = _fitters("covariance", "adam", [(64, 64, 200)])[0]
    lib = _lib.load()
    table = torch.empty(int(lib.gi2d_batch_bytes(2)), dtype=torch.uint8, device=DEV)
    states = (C.c_void_p * 2)(C.addressof(a.state), C.addressof(b.state))
    lr3 = (C.c_double * 3)(1e-3, 1e-3, 1e-3)
    rc = lib.gi2d_train_steps_batched(2, states, table.data_ptr(), table.numel(), lr3, 0.9, 0.999, 1e-8, 1, 1, None)
    assert rc != 0 and b"share model kind" in lib.gi2d_last_error_string()
    rc = lib.gi2d_train_steps_batched(2, states, table.data_ptr(), 64, lr3, 0.9, 0.999, 1e-8, 1, 1, None)
    assert rc != 0 and b"batch table" in lib.gi2d_last_error_string()
    with pytest.raises(ValueError, match="differs from fitter 0"):
        BatchFitter([a, b])
    # a table that is not 16-byte aligned, more than 64 images, no images
    c = _fitters("cholesky", "adam", [(64, 64, 150)])[0]
    two = (C.c_void_p * 2)(C.addressof(a.state), C.addressof(c.state))
    big = torch.empty(int(lib.gi2d_batch_bytes(64)) + 64, dtype=torch.uint8, device=DEV)
    rc = lib.gi2d_train_steps_batched(2, two, big.data_ptr() + 4, big.numel() - 4, lr3, 0.9, 0.999, 1e-8, 1, 1, None)
    assert rc != 0 and b"aligned" in lib.gi2d_last_error_string()
    many = (C.c_void_p * 65)(*([C.addressof(a.state)] * 65))
    assert lib.gi2d_train_steps_batched(65, many, big.data_ptr(), big.numel(), lr3, 0.9, 0.999, 1e-8, 1, 1, None) != 0
    assert lib.gi2d_train_steps_batched(0, many, big.data_ptr(), big.numel(), lr3, 0.9, 0.999, 1e-8, 1, 1, None) != 0
    assert lib.gi2d_train_steps_batched(2, two, big.data_ptr(), big.numel(), lr3, 0.9, 0.999, 1e-8, 0, 1, None) != 0  # step 0
    # count 0: nothing to do, nothing touched
    before = a.xyz.clone()
    assert lib.gi2d_train_steps_batched(2, two, big.data_ptr(), big.numel(), lr3, 0.9, 0.999, 1e-8, 1, 0, None) == 0
    torch.cuda.synchronize()
    assert torch.equal(a.xyz, before)


def test_sixty_four_images_in_one_launch():
    """The table's limit (one ballot finds a workgroup's image): 64 small images of three sizes, two iterations."""
    from gaussianimage_plus_amd.trainer import BatchFitter
    sizes = [[(48, 64, 120), (64, 48, 90), (33, 50, 60)][i % 3] for i in range(64)]
    alone, together = _fitters("covariance", "adam", sizes, track_best=False), _fitters("covariance", "adam", sizes,
                                                                                      track_best=False)
    for f in alone:
        f.train(2)
    BatchFitter(together).train(2)
    torch.cuda.synchronize()
    for i, (a, b) in enumerate(zip(alone, together)):
        _assert_same(a, b, f"image {i} of 64")


QROWS = ("qparams", "qm", "qv", "best_qparams", "qfeat")


@pytest.mark.parametrize("kind,sizes", [("covariance", MIXED), ("scale_rot", UNIFORM), ("covariance", BIG),
                                        ("scale_rot", BIG)])
def test_batched_quantised_iterations_equal_single_image_calls(kind, sizes):
    """Quantisation-aware iterations (train_iter_quantize) of K images with four launches per iteration for all of them
    == the same iterations image by image: parameters, Adam moments, the quantisers' learned values and their moments,
    the dequantised colours, render, snapshots -- bit for bit.  BIG: each image alone runs its per-gaussian kernels in
    64-lane workgroups, the batch in 256-lane ones (the quantisers' whole-array sums are formed from per-WAVE rows, so
    they do not notice).  A warm-up first: the quantisers are initialised from the data."""
    from gaussianimage_plus_amd.trainer import BatchFitter
    alone, together = _fitters(kind, "adam", sizes), _fitters(kind, "adam", sizes)
    for fs in (alone, together):
        for f in fs:
            f.train(40)
            f.enable_quantize(12, None, 6, debug_grads=False)
    for f in alone:
        for c in (1, 3, 5):
            f.train(c)
    b = BatchFitter(together)
    for c in (1, 3, 5):
        b.train(c)
    torch.cuda.synchronize()
    for i, (x, y) in enumerate(zip(alone, together)):
        _assert_same(x, y, f"{kind} image {i}")
        for nm in QROWS:
            ta, tb = getattr(x, nm), getattr(y, nm)
            assert torch.equal(ta, tb), f"{kind} image {i}: {nm} differs in {int((ta != tb).sum())} elements"
        if kind == "covariance":
            assert torch.equal(x.qrange, y.qrange), f"image {i}: log range"
        y.check_status()
        # and the quantised render (forward_quantize) of the batch-trained model is the single-image one
        assert torch.equal(x.render(), y.render())


def test_batched_quantised_launcher_equals_streams_launcher():
    """launch.fit_images_native(quantize=True): train_quantize.py's loop in batches == on one stream per image."""
    from gaussianimage_plus_amd.launch import fit_images_native, synthetic_image
    gts = [synthetic_image(96, 144, 70 + i).to(DEV) for i in range(4)]
    kw = dict(lr=0.018, kind="covariance", max_points=1800, prune_iter=50, grow_iter=100, eps=1e-15, eval_renders=1,
              quantize=True, warmup_iter=200)
    a = fit_images_native(gts, 1200, 350, batched=False, threaded=True, **kw)
    for groups in (True, 2):
        b = fit_images_native(gts, 1200, 350, batched=groups, **kw)
        for ra, rb in zip(a, b):
            assert ra["mse"] == rb["mse"] and ra["num_gaussians"] == rb["num_gaussians"]
            assert ra["psnr_decoded"] == rb["psnr_decoded"] and ra["bpp"] == rb["bpp"]


def test_best_model_error_of_a_large_image_is_the_sum_of_its_tile_errors():
    """3 450 tiles: the chunked sum behind the best-model decision (csrc/gi2d_train.hip::best_decision) against torch's."""
    f = _fitters("cholesky", "adam", LARGE[:1])[0]
    f.train(1)
    torch.cuda.synchronize()
    want = float(f.tile_sse.double().sum())
    got = float(f.best_sse[1 if f.best_sse[1] < float("inf") else 0])
    assert abs(got - want) <= 1e-5 * want, (got, want)


def _placed_fitters(scenes):
    """Cholesky-model fitters on hand-placed gaussians: (h, w, n_sparse, n_cluster) -- n_sparse small gaussians (about
    3 px, a dozen candidates per tile) spread over the image, n_cluster more piled into one tile."""
    from gaussianimage_plus_amd.launch import synthetic_image
    from gaussianimage_plus_amd.trainer import NativeFitter
    out = []
    for i, (h, w, n_sparse, n_cluster) in enumerate(scenes):
        rng = np.random.default_rng(900 + i)
        n = n_sparse + n_cluster
        u = rng.uniform(-0.98, 0.98, (n, 2))
        if n_cluster:  # inside the tile at pixel (40 .. 48, 40 .. 48)
            u[n_sparse:, 0] = (rng.uniform(40, 48, n_cluster) + 0.5) / (0.5 * w) - 1.0
            u[n_sparse:, 1] = (rng.uniform(40, 48, n_cluster) + 0.5) / (0.5 * h) - 1.0
        L = np.stack([rng.uniform(0.8, 1.6, n), rng.uniform(-0.3, 0.3, n), rng.uniform(0.8, 1.6, n)], 1)
        init = {"xyz": torch.from_numpy(np.arctanh(u).astype(np.float32)), "chol": torch.from_numpy(L.astype(np.float32)),
                "feat": torch.from_numpy(rng.uniform(0, 0.2, (n, 3)).astype(np.float32)),
                "bound": torch.tensor([0.5, 0.0, 0.5])}
        out.append(NativeFitter(synthetic_image(h, w, 70 + i).to(DEV), n, kind="cholesky", lr=1e-3, seed=5, init=init))
    return out


def _crowd(fitters, count=600):
    """Pile the first `count` gaussians of every given fitter into the tile at pixel (40 .. 48, 40 .. 48)."""
    for f in fitters:
        rng = np.random.default_rng(77)
        u = np.stack([(rng.uniform(40, 48, count) + 0.5) / (0.5 * f.w) - 1.0,
                      (rng.uniform(40, 48, count) + 0.5) / (0.5 * f.h) - 1.0], 1)
        f._xyz[:count] = torch.from_numpy(np.arctanh(u).astype(np.float32)).to(DEV)


def _swell(fitters, count=160):
    """Make the first `count` gaussians of every given fitter cover the whole image: every tile's row then holds more
    candidates than the small form stages."""
    for f in fitters:
        f._chol[:count] = torch.tensor([300.0, 0.0, 300.0], device=DEV)


def test_batched_tile_pass_forms_give_the_same_bits_and_follow_the_reported_row_sizes():
    """gi2d_batch_tile_pass_form (include/gi2d.h): the first call on a table runs the general form; its report -- the
    number of tiles with more than 128 candidates -- decides the next call's: two launches while at most one tile in
    sixteen is that full.  Eight 768x512 images with short rows take the two-launch form from the second call on.  Then
    600 gaussians of one image move into one tile BETWEEN two calls: the second launch has real work from then on, and
    the form stays (one crowded tile in 12 288).  Then 160 gaussians of that image grow to cover all its 1 536 tiles: the
    next call still runs as two launches (its answer is one call old), the call after that is back to one launch.
    Every image's state equals the single-image calls' bit for bit throughout."""
    from gaussianimage_plus_amd import _lib
    from gaussianimage_plus_amd.trainer import BatchFitter
    lib = _lib.load()
    scenes = [(512, 768, 2500, 0)] * 4 + [(768, 512, 2200, 0)] * 4
    alone, together = _placed_fitters(scenes), _placed_fitters(scenes)
    batch = BatchFitter(together)
    form = lambda b: int(lib.gi2d_batch_tile_pass_form(C.c_void_p(b.table.data_ptr())))
    assert form(batch) == 0  # nothing known about this table yet

    def run(count):
        for f in alone:
            f.train(count)
        batch.train(count)
        torch.cuda.synchronize()
        return form(batch)

    def same(tag):
        for i, (a, b) in enumerate(zip(alone, together)):
            a.check_status(), b.check_status()
            _assert_same(a, b, f"{tag}, image {i}")

    assert [run(2), run(3)] == [1, 1]  # i.e. the call of 3 iterations ran as two launches
    same("two launches, nothing for the second")
    _crowd([alone[0], together[0]])
    assert run(2) == 1  # ran as two launches (decided from the report before), reports ONE fuller tile: the form stays
    same("two launches, a crowded tile for the second")
    assert int(together[0].status[3]) >= 512, "the crowded tile is what the second launch was for"
    assert run(2) == 1
    same("two launches again")
    _swell([alone[0], together[0]])
    assert run(2) == 0  # ran as two launches, an eighth of the batch's tiles for the second: the next call takes one
    same("two launches, a whole image for the second")
    assert run(2) == 0
    same("one launch")


def test_small_batches_stay_with_one_launch():
    """Fewer than eight residency rounds of tiles: the small form has no idle slots to fill, the form stays general."""
    from gaussianimage_plus_amd import _lib
    from gaussianimage_plus_amd.trainer import BatchFitter
    lib = _lib.load()
    together = _placed_fitters([(384, 512, 2500, 0), (512, 384, 2500, 0)])
    batch = BatchFitter(together)
    for count in (2, 2):
        batch.train(count)
        torch.cuda.synchronize()
        assert int(lib.gi2d_batch_tile_pass_form(C.c_void_p(batch.table.data_ptr()))) == 0
