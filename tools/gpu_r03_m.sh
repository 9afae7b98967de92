cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03
mkdir -p $O
python -m pytest tests -m gpu -x -q 2>&1 | tail -4 > gpurun_out/r03m_pytest.txt
python tools/bench_configs.py > $O/configs.jsonl 2>/dev/null
python tools/bench_refit_forms.py > $O/refit_forms.jsonl 2>/dev/null
python tools/bench_refit_forms.py f32 > $O/refit_forms_f32.jsonl 2>/dev/null
PMC="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE"
rm -rf $O/pmc_refit_C2
rocprofv3 --pmc $PMC --kernel-trace --output-format csv -d $O/pmc_refit_C2 -- python3 tools/bench_configs.py C2 > $O/pmc_refit_C2.log 2>&1
cat gpurun_out/r03m_pytest.txt; head -1 $O/configs.jsonl | cut -c1-200
