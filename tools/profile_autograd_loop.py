"""Development aid: where the host time of the drop-in (autograd) training loop goes."""
import cProfile, pstats, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from gaussianimage_plus_amd.launch import fit_image, synthetic_image

gt = synthetic_image(512, 768, 100).cuda()
fit_image(gt, 5000, 200)
pr = cProfile.Profile()
pr.enable()
fit_image(gt, 5000, 1000)
pr.disable()
st = pstats.Stats(pr).sort_stats("cumulative")
st.print_stats(45)
