cd $GRAFT_REPO_ROOT
export BNERV_TEST_TRAIL=$PWD/gpurun_out/r05v_trail.txt
timeout 900 python -m pytest tests/test_gpu_models.py -x -q -m gpu -p no:cacheprovider -k "time_branch or tiny_models or c1_full or trajectory or reproducible or step_frame or weight_fragment or lazy_flush or short_schedule or long_unsync" > gpurun_out/r05v_models.log 2>&1; echo "models rc=$?"; grep -v "bnerv-trail" gpurun_out/r05v_models.log | tail -25 | cut -c1-300
TIMELINE=1 tools/ab_steps.sh r05v c1 "BNERV_TIME_BRANCH=1"
