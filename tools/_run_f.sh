cd $GRAFT_REPO_ROOT
export BNERV_TEST_TRAIL=$PWD/gpurun_out/r05w_trail.txt
timeout 600 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -p no:cacheprovider -k "shared_tile or tat" > gpurun_out/r05w_ops.log 2>&1; echo "ops rc=$?"; grep -v "bnerv-trail" gpurun_out/r05w_ops.log | tail -3 | cut -c1-300
timeout 600 python -m pytest tests/test_gpu_models.py -x -q -m gpu -p no:cacheprovider -k "c1_full or reproducible" > gpurun_out/r05w_models.log 2>&1; echo "models rc=$?"; grep -v "bnerv-trail" gpurun_out/r05w_models.log | tail -3 | cut -c1-300
TIMELINE=1 tools/ab_steps.sh r05w c1 "BNERV_X=1"
python tools/pmc_traffic.py pair_dk2s r05w > /dev/null 2>&1; head -4 gpurun_out/r05w_pmc_pair_dk2s.md | tail -2
