"""Development aid: where the host time of the drop-in (autograd) training loop goes -- launch.fit_image, the loop of
models/gaussianimage_cholesky.py:302-317 through the drop-in gsplat wrappers with torch Adam.
usage: profile_autograd_loop.py [N] [iterations]      (prints us per iteration, then the cProfile split)"""
import cProfile, pstats, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from gaussianimage_plus_amd.launch import fit_image, synthetic_image

n = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
gt = synthetic_image(512, 768, 100).cuda()
fit_image(gt, n, 200)
for _ in range(3):
    r = fit_image(gt, n, iters, eval_renders=1)
    print(f"N={n}: {r['train_s'] / iters * 1e6:.1f} us per iteration ({iters / r['train_s']:.0f} it/s), PSNR {r['psnr']:.2f}")
pr = cProfile.Profile()
pr.enable()
fit_image(gt, n, iters, eval_renders=1)
pr.disable()
st = pstats.Stats(pr).sort_stats("tottime")
st.print_stats(28)
