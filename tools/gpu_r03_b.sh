cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03b
mkdir -p $O
python tools/probe_ramp.py > $O/ramp.txt 2>&1
python tools/tune_jets.py run > $O/tune_jets_f32.txt 2>&1
python tools/tune_jets.py run f64 > $O/tune_jets_f64.txt 2>&1
cat $O/ramp.txt; grep -v "max rel" $O/tune_jets_f32.txt | tail -14; grep "max rel" $O/tune_jets_f32.txt | head -12; grep -v "max rel" $O/tune_jets_f64.txt | tail -14
