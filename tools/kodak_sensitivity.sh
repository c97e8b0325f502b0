#!/bin/bash
# DESIGN.md section 6 "run-to-run sensitivity": kodim10 / kodim24 / kodim01 fitted alone with the round-3, round-4 and current
# libraries, and with the round-4 library from six starts that differ by one rounding-level change.
#   (build/old_r3, build/old_r4: `git archive <round's last commit> gaussianimage_plus_amd include | tar -x`, `make -C csrc`)
cd ${GRAFT_REPO_ROOT:-.}
OUT=gpurun_out/kodak_sensitivity.jsonl
: > $OUT
IM=${IMAGES:-kodim10,kodim24,kodim01}
for root in build/old_r3 build/old_r4 .; do timeout -k 10 300 python tools/kodak_sensitivity.py $root $IM 0 >> $OUT 2>> gpurun_out/kodak_sensitivity.err; done
for p in 1 2 3 4 5 6; do timeout -k 10 300 python tools/kodak_sensitivity.py build/old_r4 $IM $p >> $OUT 2>> gpurun_out/kodak_sensitivity.err; done
timeout -k 10 300 python tools/kodak_sensitivity.py . $IM 1 >> $OUT 2>> gpurun_out/kodak_sensitivity.err
python - <<'PY'
import json
rows = [json.loads(l) for l in open("gpurun_out/kodak_sensitivity.jsonl") if l.startswith("{")]
base = {r["image"]: r for r in rows if r["root"] == "build/old_r4" and r["perturbation"] == 0}
for r in rows:
    b = base[r["image"]]["population_after_each_prune_check"]
    t = r["population_after_each_prune_check"]
    first = next((100 * (i + 1) for i, (x, y) in enumerate(zip(b, t)) if x != y), None)
    print(f"{r['root']:14s} {r['image']} perturbation {r['perturbation']}: PSNR {r['psnr']:.3f} best_n {r['best_n']} final_n {r['final_n']}"
          f"  first population difference from round-4 unperturbed at iteration {first}")
PY
