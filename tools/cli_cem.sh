#!/usr/bin/env bash
# End-to-end run of the compression pipeline (scripts/compression/hnerv_boost.sh flags) on a short synthetic UVG-shaped clip:
# 1) a few regression epochs with train_nerv_all.py to get a checkpoint, 2) train_nerv_compression.py from that checkpoint.
# usage: tools/cli_cem.sh [frames] [regression epochs] [compression epochs]
N=${1:-16}; E1=${2:-3}; E2=${3:-3}
cd "$(dirname "$0")/.."
rm -rf output/cem_reg output/cem_cmp
COMMON="--model HNeRV_Boost --sft_block res_sft --ch_t 32 --data_path synthetic:${N}x1080x1920 --vid synth --optim_type Adan --conv_type convnext pshuffel_3x3 --act sin --norm none --crop_list 1080_1920 --resize_list -1 --loss Fusion10_freq --embed pe_1.25_80 --enc_strds 5 3 2 2 2 --enc_dim 64_16 --dec_strds 5 3 2 2 2 --ks 0_1_5 --reduce 1.2 --dec_blks 1 1 2 2 2 --modelsize 1.0 --lower_width 12 -b 1"
python train_nerv_all.py --outf cem_reg $COMMON -e $E1 --eval_freq $E1 --lr 0.003 2>&1 | grep -E "Eval at epoch|Training wo|Error|Traceback" | cut -c1-220 | tail -3
python train_nerv_compression.py --outf cem_cmp $COMMON -e $E2 --eval_freq $E2 --lr 0.0005 --weight output/cem_reg/synth/Size1.0/model_latest.pth \
  --lr_type cosine_0_1_0.1 --not_resume --embed_entropy --quant --quant_model_bit 8 --quant_bias_bit 8 --quant_embed_bit 8 --quantizer_w scale \
  --quantizer_b scale --quantizer_e scalebeta --lambda_rate 0.05 --target_bit 4 2>&1 | grep -E "Epoch\[|Time/epoch|Eval at epoch|Gaussian Entropy|Training|Error|Traceback|rror" | cut -c1-260 | tail -14
