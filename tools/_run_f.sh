cd $GRAFT_REPO_ROOT
export BNERV_TEST_TRAIL=$PWD/gpurun_out/r05j_trail.txt
BNERV_PAIR_FUSED=8 timeout 600 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -p no:cacheprovider -k "tat or snerv or blocks or low_res or wgrad or conv" > gpurun_out/r05j_ops.log 2>&1; echo "ops rc=$?"; grep -v "bnerv-trail" gpurun_out/r05j_ops.log | tail -3 | cut -c1-300
BNERV_PAIR_FUSED=8 timeout 600 python -m pytest tests/test_gpu_models.py -x -q -m gpu -p no:cacheprovider -k "tiny_models or c1_full or trajectory or reproducible" > gpurun_out/r05j_models.log 2>&1; echo "models rc=$?"; grep -v "bnerv-trail" gpurun_out/r05j_models.log | tail -3 | cut -c1-300
TIMELINE=1 tools/ab_steps.sh r05j c1 "BNERV_PAIR_FUSED=0" "BNERV_PAIR_FUSED=2000"
