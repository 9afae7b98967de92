# A/B sweeps of the posterior kernel's tuning macros on the GPU box:  bash tools/dev/sweep_posterior_flags.sh "<flags>" ...
for flags in "$@"; do
  touch bayesian_cbf_amd/csrc/posterior_step.hip
  BCBF_EXTRA_HIPCC_FLAGS="$flags" python -m bayesian_cbf_amd.build > /dev/null 2>&1 || { echo "$flags BUILD FAILED"; continue; }
  python bench.py --cpu-sample 0 --steps 100 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(sys.argv[1], round(d['value']), round(d['ms_per_step'],4), round(d['roofline']['kernel_ms'],4), round(d['roofline']['frac'],4))" "[$flags]"
done
