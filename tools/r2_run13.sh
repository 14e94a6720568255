cd $GRAFT_REPO_ROOT
timeout -k 10 800 python -m pytest tests -q -m gpu -x > gpurun_out/r2_run13_tests.log 2>&1; echo rc=$?; tail -8 gpurun_out/r2_run13_tests.log | cut -c1-300
for pt in 128 64; do echo "--- predict tile $pt"; GPRX_PREDICT_TILE=$pt timeout -k 10 400 python bench.py --steps 5 --warmup 2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); e=d.get('extra',{})
print({k:e.get(k) for k in ('F2_objective_grad_evals_per_s','F2_batched_objective_grad_evals_per_s','F2_batched_tflops','predict_points_per_s','predict_tflops','C4_fit_plus_predict_100k_cells_per_s','F3_lbfgs50_matern52_ard_seconds','F3_lockstep_16_modes_seconds')}, d.get('extra_error'), d.get('parity_at_bench_size'))"; done
