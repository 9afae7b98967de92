# A/B / ablation sweeps of the fp64 refit kernel on the GPU box:  bash tools/dev/sweep_refit64_flags.sh "<flags>" ...
for flags in "$@"; do
  touch bayesian_cbf_amd/csrc/refit_mfma64.hip
  BCBF_EXTRA_HIPCC_FLAGS="$flags" python -m bayesian_cbf_amd.build > /dev/null 2>&1 || { echo "$flags BUILD FAILED"; continue; }
  BCBF_ABLATION=1 python tools/bench_configs.py C2 C3f64 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(sys.argv[1], d['config'], 'refit_ms', round(d['refit_ms'], 3), 'TF', round(d['refit_TFLOPs'], 1))" "[$flags]" 2>&1 | tail -2
done
