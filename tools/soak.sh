#!/bin/bash
# End-to-end soak of the launcher (one GPU): every model / optimizer / schedule the reference's scripts wire, on several
# synthetic images of two sizes.  Each line must end with an "Average:" row and a plausible PSNR.
cd ${GRAFT_REPO_ROOT:-/root/repo}
export PYTHONPATH=$PWD:$PYTHONPATH
run() { echo "== $*"; timeout -k 10 280 python -m gaussianimage_plus_amd.launch "$@" 2>&1 | grep -E "Average|Error|error|Traceback|overflow" | cut -c1-220; }
run --synthetic 6 --model covariance --num_points 5000 --max_num_points 50000 --iterations 8000 --grow_iter 800
run --synthetic 4 --model covariance --num_points 2500 --max_num_points 5000 --iterations 6000 --grow_iter 500 --quantize --warmup_iter 2000
run --synthetic 4 --model scale_rot --num_points 30000 --iterations 3000 --quantize --warmup_iter 1000
run --synthetic 4 --model cholesky --num_points 10000 --iterations 3000
run --synthetic 4 --model cholesky --num_points 10000 --iterations 3000 --opt_type adam
run --synthetic 2 --model covariance --num_points 20000 --max_num_points 60000 --iterations 3000 --grow_iter 300 --height 1356 --width 2040
run --synthetic 3 --model covariance --num_points 3000 --iterations 300 --loop autograd
run --synthetic 3 --model cholesky --num_points 3000 --iterations 600 --loop autograd --graph
run --synthetic 5 --model covariance --num_points 5000 --max_num_points 50000 --iterations 4000 --grow_iter 400 --height 500 --width 333
# several images per GPU: three batches on three streams (the default), one stream per image, quantised in batches
run --synthetic 6 --model covariance --num_points 2500 --max_num_points 20000 --iterations 4000 --grow_iter 400 --images_per_gpu 6
run --synthetic 6 --model covariance --num_points 2500 --max_num_points 20000 --iterations 4000 --grow_iter 400 --images_per_gpu 6 --streams
run --synthetic 5 --model covariance --num_points 2500 --max_num_points 5000 --iterations 6000 --grow_iter 500 --quantize --warmup_iter 2000 --images_per_gpu 5
run --synthetic 4 --model scale_rot --num_points 30000 --iterations 3000 --quantize --warmup_iter 1000 --images_per_gpu 4 --batch_groups 2
