#!/usr/bin/env python3
"""Kodak-24 as a data fixture (run in the DEV container only; writes kodak24.npz).

The reference fits the 24 images of datasets/kodak (kodim01..24.png: 18 landscape 768x512, 6 portrait 512x768) one
after the other (train.py:294-308); BASELINE.json's second metric -- Kodak images/sec -- is quoted on exactly that set.
/root/reference does not exist on the GPU box, so the decoded pixels travel as data: one uint8 [H, W, 3] array per image
(the bytes PIL hands to the reference's own loader, utils.py:21-26 image_path_to_tensor: Image.open -> ToTensor = /255).
No reference source is stored, only pixels."""
import glob
import os

import numpy as np
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = "/root/reference/datasets/kodak"


def main():
    out = {}
    for f in sorted(glob.glob(os.path.join(SRC, "kodim*.png"))):
        a = np.asarray(Image.open(f).convert("RGB"), dtype=np.uint8)
        assert a.shape in ((512, 768, 3), (768, 512, 3)), (f, a.shape)
        out[os.path.splitext(os.path.basename(f))[0]] = a
    assert len(out) == 24
    portrait = sum(1 for a in out.values() if a.shape[0] > a.shape[1])
    assert portrait == 6, portrait
    np.savez_compressed(os.path.join(HERE, "kodak24.npz"), **out)
    print({k: v.shape for k, v in out.items()})
    print(os.path.getsize(os.path.join(HERE, "kodak24.npz")) / 2 ** 20, "MiB")


if __name__ == "__main__":
    main()
