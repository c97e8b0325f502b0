"""Training iterations of K images per launch (gi2d_train_steps_batched) against the same images fitted one launch
each: microseconds per image-iteration.  Usage: batch_time.py [N] [H] [W] [kind] [K ...]   (development aid)"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gaussianimage_plus_amd.launch import synthetic_image  # noqa: E402
from gaussianimage_plus_amd.trainer import BatchFitter, NativeFitter  # noqa: E402

dev = torch.device("cuda:0")


def main():
    a = sys.argv[1:]
    n, h, w = int(a[0]) if a else 50000, int(a[1]) if len(a) > 1 else 512, int(a[2]) if len(a) > 2 else 768
    kind = a[3] if len(a) > 3 else "cholesky"
    ks = [int(x) for x in a[4:]] or [1, 2, 4, 8, 24]
    iters = 400
    for k in ks:
        fit = [NativeFitter(synthetic_image(h, w, 100 + i).to(dev), n, kind=kind, lr=1e-3, seed=3047 + i,
                            track_best=True) for i in range(k)]
        b = BatchFitter(fit)
        b.train(50)
        torch.cuda.synchronize()
        t0 = time.time()
        b.train(iters)
        torch.cuda.synchronize()
        dt = time.time() - t0
        print(f"K={k:2d} {kind} N={n} {w}x{h}: {dt / iters * 1e6:8.1f} us per batch iteration, "
              f"{dt / iters / k * 1e6:7.2f} us per image-iteration, {k * iters / dt:9.0f} image-iterations/s", flush=True)
        del b, fit
        torch.cuda.empty_cache()
        torch.cuda.synchronize()
        time.sleep(0.5)  # returning gigabytes to the driver stalls the queue for ~0.1 s: keep it out of the next timing
    fit = NativeFitter(synthetic_image(h, w, 100).to(dev), n, kind=kind, lr=1e-3, seed=3047, track_best=True)
    fit.train(50)
    torch.cuda.synchronize()
    t0 = time.time()
    fit.train(iters)
    torch.cuda.synchronize()
    dt = time.time() - t0
    print(f"single-image calls: {dt / iters * 1e6:.1f} us per iteration", flush=True)


if __name__ == "__main__" and not os.environ.get("GI2D_STREAMS"):
    main()


def streams(n, h, w, kind, k, iters=400):
    """The round-2 form: K images on K HIP streams, one host thread each (for comparison with the batched launch)."""
    import threading
    fits = [NativeFitter(synthetic_image(h, w, 100 + i).to(dev), n, kind=kind, lr=1e-3, seed=3047 + i, track_best=True)
            for i in range(k)]
    sts = [torch.cuda.Stream(device=dev) for _ in fits]

    def drive(i, count):
        with torch.cuda.device(dev), torch.cuda.stream(sts[i]):
            fits[i].train(count)

    def run(count):
        ts = [threading.Thread(target=drive, args=(i, count)) for i in range(k)]
        for t in ts:
            t.start()
        for t in ts:
            t.join()
        torch.cuda.synchronize()
    run(50)
    t0 = time.time()
    run(iters)
    dt = time.time() - t0
    print(f"K={k:2d} streams+threads {kind} N={n} {w}x{h}: {dt / iters / k * 1e6:7.2f} us per image-iteration, "
          f"{k * iters / dt:9.0f} image-iterations/s", flush=True)


if __name__ == "__main__" and os.environ.get("GI2D_STREAMS"):
    a = sys.argv[1:]
    for k in [int(x) for x in a[4:]] or [4, 8, 24]:
        streams(int(a[0]) if a else 50000, int(a[1]) if len(a) > 1 else 512, int(a[2]) if len(a) > 2 else 768,
                a[3] if len(a) > 3 else "cholesky", k)
