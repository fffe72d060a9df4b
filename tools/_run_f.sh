cd $GRAFT_REPO_ROOT
export BNERV_TEST_TRAIL=$PWD/gpurun_out/r05z_trail.txt
timeout 600 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -p no:cacheprovider -k "head" > gpurun_out/r05z_ops.log 2>&1; echo "ops rc=$?"; grep -v "bnerv-trail" gpurun_out/r05z_ops.log | tail -3 | cut -c1-300
timeout 600 python -m pytest tests/test_gpu_models.py -x -q -m gpu -p no:cacheprovider -k "tiny_models or c1_full or reproducible or trajectory or decode or big_models_full" > gpurun_out/r05z_models.log 2>&1; echo "models rc=$?"; grep -v "bnerv-trail" gpurun_out/r05z_models.log | tail -3 | cut -c1-300
TIMELINE=1 tools/ab_steps.sh r05z c1 "BNERV_HEAD_FWD=0" "BNERV_HEAD_FWD=1"
