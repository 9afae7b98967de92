cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03
mkdir -p $O
python -m pytest tests -m gpu -x -q 2>&1 | tail -4 > gpurun_out/r03n_pytest.txt
timeout 300 python tools/check_refit_forms.py 2>&1 | tail -8
python tools/bench_configs.py > $O/configs.jsonl 2>/dev/null
python tools/bench_refit_forms.py > $O/refit_forms.jsonl 2>/dev/null
python tools/bench_refit_forms.py f32 > $O/refit_forms_f32.jsonl 2>/dev/null
cat gpurun_out/r03n_pytest.txt; cut -c1-330 $O/configs.jsonl
