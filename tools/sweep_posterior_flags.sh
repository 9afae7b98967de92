for flags in "-DBCBF_PS_WAVES=3" "-DBCBF_PS_WAVES=4" "-DBCBF_PS_UNR=8" "-DBCBF_PS_UNR=8 -DBCBF_PS_WAVES=3" "-DBCBF_PS_UNR=2 -DBCBF_PS_WAVES=4" "-DBCBF_PS_PKASM=0"; do
  touch bayesian_cbf_amd/csrc/posterior_step.hip
  BCBF_EXTRA_HIPCC_FLAGS="$flags" python -m bayesian_cbf_amd.build > /dev/null 2>&1 || { echo "$flags BUILD FAILED"; continue; }
  python bench.py --cpu-sample 0 --steps 100 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(sys.argv[1], round(d['value']), round(d['ms_per_step'],4), round(d['roofline']['kernel_ms'],4), round(d['roofline']['frac'],4))" "$flags"
done
