#!/usr/bin/env bash
# Round-2 evidence on the GPU box -> gpurun_out/e2_* (bench lines, kernel traces, per-launch timelines, PMC passes, micro-benchmarks).
# usage (through gpurun, from the repo root): tools/round2_evidence.sh
set -u
R=$PWD
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python $R/bench.py > $O/e2_bench_c1.json 2> $O/e2_bench_c1.err
for c in c3 c4 c5; do python $R/bench.py --config $c 2>/dev/null | tail -1 > $O/e2_bench_$c.json; done
# kernel traces (--stats) of the graph-replayed step for c1 / c3 / c4, and an eager per-launch timeline for c1 / c3
rocprofv3 --kernel-trace --stats -d /tmp/ks_c1 -- python $R/bench.py --steps 40 --warmup 6 --no_cpu_baseline --no_graph > /tmp/ks_c1.log 2>&1
python $R/tools/prof_summary.py /tmp/ks_c1 46 45 > $O/e2_c1_trace.md 2>&1
for c in c3 c4; do
  rocprofv3 --kernel-trace --stats -d /tmp/ks_$c -- python $R/bench.py --config $c --steps 20 --warmup 5 --no_cpu_baseline > /tmp/ks_$c.log 2>&1
  python $R/tools/prof_summary.py /tmp/ks_$c 25 40 > $O/e2_${c}_trace.md 2>&1
done
for c in c1 c3; do
  rocprofv3 --kernel-trace --output-format csv -d /tmp/kt_$c -- python $R/bench.py --config $c --steps 4 --warmup 5 --no_cpu_baseline --no_graph > /tmp/kt_$c.log 2>&1
  python $R/tools/ktimeline.py /tmp/kt_$c loss_final_kernel > $O/e2_timeline_$c.md 2>&1
done
# PMC of the step's dominant instantiation (K2s: conv_lean_kernel<3,1,7,3>) and of the lean weight gradient
$R/tools/pmc_bf.sh conv_k2s "conv_lean_kernel<3, 1, 7" $O/e2_pmc_k2s.txt
$R/tools/pmc_bf.sh wgrad wgrad_lean_kernel $O/e2_pmc_wgrad.txt
# micro-benchmarks: f32 kernels, the three split modes, the issue / accuracy ubenches
python $R/tools/kbench.py 30 > $O/e2_kbench.txt 2>&1
for m in bf16x6 f16x3 bf16x3; do BNERV_SPLIT=$m python $R/tools/kbench.py 30 2>/dev/null | head -19 > $O/e2_kbench_$m.txt; done
BNERV_SPLIT=bf16x6 $R/tools/pmc_bf.sh conv conv_bf_kernel $O/e2_pmc_bf16x6.txt
for u in mfma_interleave bf16_split; do [ -x $R/tools/ubench/$u ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o $R/tools/ubench/$u $R/tools/ubench/$u.cpp; done   # (binaries are not tracked)
$R/tools/ubench/mfma_interleave > $O/e2_ub_interleave.txt 2>&1
$R/tools/ubench/bf16_split > $O/e2_ub_bf16split.txt 2>&1
# wide split kernels (default on): conv and weight gradient at the C3 / C4 shapes against the f32 kernels, and their PMC passes
python $R/tools/kwide2.py 20 > $O/e2_kwide_on.txt 2>&1
BNERV_SPLIT_WIDE=off python $R/tools/kwide2.py 20 > $O/e2_kwide_off.txt 2>&1
python $R/tools/kwgrad2.py 20 > $O/e2_kwgrad_on.txt 2>&1
BNERV_SPLIT_WIDE=off python $R/tools/kwgrad2.py 20 > $O/e2_kwgrad_off.txt 2>&1
$R/tools/pmc_bf.sh conv38_k2s conv_bfw_kernel $O/e2_pmc_bfw.txt
$R/tools/pmc_bf.sh wgrad38 wgrad_bfw_kernel $O/e2_pmc_wbfw.txt
echo done
