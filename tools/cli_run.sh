#!/usr/bin/env bash
# End-to-end run of the reference CLI (scripts/regression/bunny/nerv_boost.sh flags) on the synthetic Bunny-shaped clip:
# a few epochs of training + the evaluation / quantisation / Huffman reporting.  usage: tools/cli_run.sh [epochs] [extra flags]
E=${1:-3}; shift || true
cd "$(dirname "$0")/.."
rm -rf output/cli_run
python train_nerv_all.py --outf cli_run --data_path synthetic:bunny --vid bunny --model NeRV_Boost --sft_block res_sft --ch_t 32 \
  --optim_type Adan --conv_type convnext pshuffel_3x3 --act sin --norm none --crop_list 720_1280 --resize_list -1 --loss Fusion10_freq \
  --embed pe_1.25_80 --fc_hw 9_16 --dec_strds 5 2 2 2 2 --ks 0_3_3 --reduce 2 --dec_blks 1 1 2 2 2 --modelsize 0.8 -e $E --eval_freq $E \
  --lower_width 12 -b 1 --lr 0.003 "$@" 2>&1 | grep -E "Epoch|Eval|eval|Train|PSNR|psnr|bpp|time|Time|fps|FPS|Error|error|Traceback" | tail -25
