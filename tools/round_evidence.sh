#!/usr/bin/env bash
# Collects the per-round evidence on the GPU box into gpurun_out/$TAG_* (bench lines, rocprofv3 kernel traces, PMC passes,
# micro-benchmarks).  usage: tools/round_evidence.sh TAG        (run through gpurun from the repo root)
set -u
TAG=${1:-x}
R=$PWD
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python $R/bench.py > $O/${TAG}_bench_c1.json 2> $O/${TAG}_bench_c1.err
for c in c3 c4 c5; do python $R/bench.py --config $c --no_cpu_baseline 2>/dev/null | tail -1 > $O/${TAG}_bench_$c.json; done
rocprofv3 --kernel-trace --stats -d /tmp/kt_c1 -- python $R/bench.py --steps 40 --warmup 6 --no_cpu_baseline --no_graph > /tmp/kt_c1.log 2>&1
python $R/tools/prof_summary.py /tmp/kt_c1 46 45 > $O/${TAG}_c1_trace.md 2>&1
for c in c3 c4; do
  rocprofv3 --kernel-trace --stats -d /tmp/kt_$c -- python $R/bench.py --config $c --steps 20 --warmup 5 --no_cpu_baseline > /tmp/kt_$c.log 2>&1
  python $R/tools/prof_summary.py /tmp/kt_$c 25 40 > $O/${TAG}_${c}_trace.md 2>&1
done
for k in conv wgrad conv38 wgrad38; do
  i=0
  for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"; do
    i=$((i+1))
    rocprofv3 --kernel-trace --output-format csv --pmc $grp -d /tmp/pmc_$k/pmc_${k}_$i -- python $R/tools/kone.py $k 4 > /dev/null 2>&1
  done
done
: > $O/${TAG}_pmc.txt
python $R/tools/pmc_parse.py /tmp/pmc_conv conv_lean_kernel >> $O/${TAG}_pmc.txt 2>&1
python $R/tools/pmc_parse.py /tmp/pmc_wgrad wgrad_lean_kernel >> $O/${TAG}_pmc.txt 2>&1
python $R/tools/pmc_parse.py /tmp/pmc_conv38 conv_lean2_kernel >> $O/${TAG}_pmc.txt 2>&1
python $R/tools/pmc_parse.py /tmp/pmc_wgrad38 wgrad_wide_kernel >> $O/${TAG}_pmc.txt 2>&1
python $R/tools/kbench.py 30 > $O/${TAG}_kbench.txt 2>&1
python $R/tools/klean2.py 2>&1 | grep -v amdgpu.ids > $O/${TAG}_klean2.txt
python $R/tools/kwide.py 2>&1 | grep -v amdgpu.ids > $O/${TAG}_kwide.txt
echo done
