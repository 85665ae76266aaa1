cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/ab
{
timeout 3000 python -m pytest tests -q -m gpu 2>&1 | tail -12
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3
} > gpurun_out/ab/tests_full.txt 2>&1
cat gpurun_out/ab/tests_full.txt
