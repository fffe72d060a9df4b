"""Train / eval driver -- drop-in for the reference's train_nerv_all.py: every flag of its CLI (train_nerv_all.py:28-110)
with the same defaults, the same derived fields (fc_dim solver :193-217, exp_id/outf :118-137), the same train loop
semantics (:322-410) and evaluate() metrics (:451-619), so scripts/regression/**.sh run unchanged.  Differences, all
MI355X-motivated and listed in DESIGN.md:
  * the step runs through engine.TrainStep (HIP kernels, hipGraph replay, device-side PSNR, no per-step host sync);
  * data parallelism is one flat-bucket RCCL all-reduce per step (dp.GradBucket) instead of DistributedDataParallel;
    launch with -d as the reference (mp.spawn) or under torchrun (RANK/WORLD_SIZE in the environment);
  * `--data_path synthetic:bunny|uvg|NxHxW` selects the in-repo synthetic clip; frames are kept resident in HBM;
  * evaluation metrics of all ranks are combined correctly (the reference drops the all-reduce result, :554-556).
TensorBoard, GIF dumps and the seaborn histogram branch are reporting-only and optional/absent."""
import argparse
import copy
import heapq
import os
import random
import shutil
import time

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.optim as optim
import torch.utils.data
from torch.utils.data import Subset

from . import ops
from . import runtime as rt
from .dp import GradBucket
from .engine import TrainStep
from .hnerv_utils import TransformInput, VideoDataSet, adjust_lr, data_split, loss_fn, psnr_fn_device, quant_tensor, worker_init_fn
from .model_enerv import ENeRV_Boost
from .model_hnerv import HNeRV, HNeRV_Boost
from .model_nerv import NeRV_Boost


def build_parser():
    parser = argparse.ArgumentParser()
    # Dataset parameters
    parser.add_argument('--data_path', type=str, default='', help='data path for vid (or synthetic:bunny|uvg|NxHxW)')
    parser.add_argument('--vid', type=str, default='k400_train0', help='video id')
    parser.add_argument('--shuffle_data', action='store_true', help='randomly shuffle the frame idx')
    parser.add_argument('--data_split', type=str, default='1_1_1', help='Valid_train/total_train/all data split')
    parser.add_argument('--crop_list', type=str, default='640_1280', help='video crop size')
    parser.add_argument('--resize_list', type=str, default='-1', help='video resize size')
    # architecture
    parser.add_argument('--model', type=str, default='', help='model name')
    parser.add_argument('--embed', type=str, default='', help='empty string for HNeRV, and base value/embed_length for NeRV position encoding')
    parser.add_argument('--ks', type=str, default='0_3_3', help='kernel size for encoder and decoder')
    parser.add_argument('--enc_blks', type=int, default=1, help='the number of encoder blocks')
    parser.add_argument('--enc_strds', type=int, nargs='+', default=[], help='stride list for encoder')
    parser.add_argument('--enc_dim', type=str, default='64_16', help='enc latent dim and embedding ratio')
    parser.add_argument('--modelsize', type=float, default=1.5, help='model parameters size: model size + embedding parameters')
    parser.add_argument('--saturate_stages', type=int, default=-1, help='saturate stages for model size computation')
    parser.add_argument('--lfreq', type=str, default="pi", help='frequency multiplier of the positional encoding')
    parser.add_argument('--fc_dim', type=int, default=None, help='channel width of the stem output')
    parser.add_argument('--fc_hw', type=str, default='9_16', help='out size (h,w) for mlp')
    parser.add_argument('--reduce', type=float, default=1.2, help='chanel reduction for next stage')
    parser.add_argument('--lower_width', type=int, default=32, help='lowest channel width for output feature maps')
    parser.add_argument('--dec_strds', type=int, nargs='+', default=[5, 3, 2, 2, 2], help='strides list for decoder')
    parser.add_argument('--dec_blks', type=int, nargs='+', default=[1, 1, 1, 1, 1], help='block number for decoder')
    parser.add_argument("--conv_type", default=['convnext', 'pshuffel'], type=str, nargs="+", help='conv type for encoder/decoder',
                        choices=['pshuffel', 'conv', 'convnext', 'interpolate', 'pshuffel_3x3'])
    parser.add_argument('--norm', default='none', type=str, help='norm layer for generator', choices=['none', 'bn', 'in'])
    parser.add_argument('--act', type=str, default='gelu', help='activation to use',
                        choices=['relu', 'leaky', 'leaky01', 'relu6', 'gelu', 'swish', 'softplus', 'hardswish', 'sin', 'ressin'])
    parser.add_argument('--sft_block', type=str, default='none', help='TAT block type')
    parser.add_argument('--ch_t', type=int, default=32, help='sft in channels')
    parser.add_argument('--block_dim', type=int, default=128, help='transformer dims in the enerv model')
    # training
    parser.add_argument('-j', '--workers', type=int, help='number of data loading workers', default=4)
    parser.add_argument('-b', '--batchSize', type=int, default=1, help='input batch size')
    parser.add_argument('--start_epoch', type=int, default=-1, help='starting epoch')
    parser.add_argument('--not_resume', action='store_true', help='not resume from latest checkpoint')
    parser.add_argument('-e', '--epochs', type=int, default=5, help='Epoch number')
    parser.add_argument('--block_params', type=str, default='1_1', help='residual blocks and percentile to save')
    parser.add_argument('--lr', type=float, default=0.001, help='learning rate')
    parser.add_argument('--lr_type', type=str, default='cosine_0.1_1_0.1', help='learning rate type')
    parser.add_argument('--loss', type=str, default='Fusion6', help='loss type')
    parser.add_argument('--out_bias', default='tanh', type=str, help='using sigmoid/tanh/0.5 for output prediction')
    parser.add_argument('--optim_type', default='adan', type=str, help='Adan | Adam')
    parser.add_argument('--clip_max_norm', default=0., type=float, help='clip_max_norm')
    parser.add_argument('--inpanting', default='none', type=str, help='do inpanting')
    parser.add_argument('--interpolation', action='store_true', default=False, help='do interpolation')
    parser.add_argument('--embed_inter', action='store_true', default=False, help='do interpolation')
    parser.add_argument('--cabac', action='store_true', default=False)
    # evaluation
    parser.add_argument('--quant', action='store_true', default=False, help='enable quantization')
    parser.add_argument('--eval_only', action='store_true', default=False, help='do evaluation only')
    parser.add_argument('--eval_freq', type=int, default=10, help='evaluation frequency')
    parser.add_argument('--quant_model_bit', type=int, default=8, help='bit length for model quantization')
    parser.add_argument('--quant_embed_bit', type=int, default=6, help='bit length for embedding quantization')
    parser.add_argument('--quant_axis', type=int, default=0, help='quantization axis (-1 means per tensor)')
    parser.add_argument('--dump_images', action='store_true', default=False, help='dump the prediction images')
    parser.add_argument('--dump_videos', action='store_true', default=False, help='concat the prediction images into video')
    parser.add_argument('--eval_fps', action='store_true', default=False, help='fwd multiple times to test the fps')
    parser.add_argument('--encoder_file', default='', type=str, help='specify the embedding file')
    parser.add_argument('--dump_values', action='store_true', default=False)
    parser.add_argument('--dump_features', action='store_true', default=False)
    # distributed
    parser.add_argument('--manualSeed', type=int, default=1, help='manual seed')
    parser.add_argument('-d', '--distributed', action='store_true', default=False, help='distributed training')
    # logging
    parser.add_argument('--debug', action='store_true', help='debug status, earlier for train/eval')
    parser.add_argument('-p', '--print-freq', default=50, type=int)
    parser.add_argument('--weight', default='None', type=str, help='pretrained weights for ininitialization')
    parser.add_argument('--overwrite', action='store_true', help='overwrite the output dir if already exists')
    parser.add_argument('--outf', default='unify', help='folder to output images and model checkpoints')
    parser.add_argument('--suffix', default='', help="suffix str for outf")
    # MI355X-side switches (not in the reference)
    parser.add_argument('--no_graph', action='store_true', help='disable hipGraph replay of the train step')
    parser.add_argument('--host_frames', action='store_true', help='stream frames through the DataLoader instead of keeping them in HBM')
    return parser


def solve_fc_dim(args, final_size, full_data_length):
    """The reference's inline size solver (train_nerv_all.py:193-217): channel width of the stem such that decoder params +
    embedding params ~= --modelsize.  Returns (fc_dim, embed_param) and rewrites args.enc_dim as the reference does."""
    if ('pe' in args.embed or 'le' in args.embed) and "HNeRV_Boost" not in args.model:
        embed_param = 0
        embed_dim = int(args.embed.split('_')[-1]) * 2
        fc_param = np.prod([int(x) for x in args.fc_hw.split('_')])
    else:
        total_enc_strds = np.prod(args.enc_strds)
        embed_hw = final_size / total_enc_strds ** 2
        enc_dim1, embed_ratio = [float(x) for x in args.enc_dim.split('_')]
        embed_dim = int(embed_ratio * args.modelsize * 1e6 / full_data_length / embed_hw) if embed_ratio < 1 else int(embed_ratio)
        embed_param = float(embed_dim) / total_enc_strds ** 2 * final_size * full_data_length
        if args.interpolation:
            embed_param = embed_param / 2
        args.enc_dim = f'{int(enc_dim1)}_{embed_dim}'
        fc_param = (np.prod(args.enc_strds) // np.prod(args.dec_strds)) ** 2 * 9
    decoder_size = args.modelsize * 1e6 - embed_param
    ch_reduce = 1. / args.reduce
    dec_ks1, dec_ks2 = [int(x) for x in args.ks.split('_')[1:]]
    fix_ch_stages = len(args.dec_strds) if args.saturate_stages == -1 else args.saturate_stages
    a = ch_reduce * sum([ch_reduce ** (2 * i) * s ** 2 * min((2 * i + dec_ks1), dec_ks2) ** 2 for i, s in enumerate(args.dec_strds[:fix_ch_stages])])
    b = embed_dim * fc_param
    c = args.lower_width ** 2 * sum([s ** 2 * min(2 * (fix_ch_stages + i) + dec_ks1, dec_ks2) ** 2 for i, s in enumerate(args.dec_strds[fix_ch_stages:])])
    fc_dim = args.fc_dim if args.fc_dim is not None else int(np.roots([a, b, c - decoder_size]).max())
    return fc_dim, embed_param


def build_model(args):
    if args.model == "NeRV_Boost":
        args.expansion = 1
        return NeRV_Boost(1, args=args)
    if args.model == "ENeRV_Boost":
        args.expansion = 3
        return ENeRV_Boost(3, args=args)
    if args.model == "HNeRV_Boost":
        return HNeRV_Boost(args)
    if args.model == "HNeRV":
        return HNeRV(args)
    raise ValueError(f"unknown --model {args.model!r}")


def main(argv=None):
    args = build_parser().parse_args(argv)
    torch.set_printoptions(precision=2)
    if args.debug:
        args.eval_freq, args.outf = 1, 'output/debug'
    else:
        args.outf = os.path.join('output', args.outf)
    args.enc_strd_str = ','.join(str(x) for x in args.enc_strds)
    args.dec_strd_str = ','.join(str(x) for x in args.dec_strds)
    args.quant_str = f'quant_M{args.quant_model_bit}_E{args.quant_embed_bit}'
    args.exp_id = f'{args.vid}/Size{args.modelsize}'
    args.outf = os.path.join(args.outf, args.exp_id)
    if args.overwrite and os.path.isdir(args.outf):
        print('Will overwrite the existing output dir!')
        shutil.rmtree(args.outf)
    os.makedirs(args.outf, exist_ok=True)
    args.ngpus_per_node = torch.cuda.device_count()
    env_world = int(os.environ.get("WORLD_SIZE", "1"))
    if env_world > 1:                                      # launched by torchrun: one process per GPU exists already
        args.distributed, args.ngpus_per_node, args.init_method = True, env_world, "env://"
        train(int(os.environ.get("LOCAL_RANK", 0)), args)
    elif args.distributed and args.ngpus_per_node > 1:     # the reference's own launch form: -d spawns one process per GPU
        args.init_method = f'tcp://127.0.0.1:{hash(args.exp_id) % 20000 + 10000}'
        print(f'init_method: {args.init_method}', flush=True)
        mp.spawn(train, nprocs=args.ngpus_per_node, args=(args,))
    else:
        train(None, args)


def data_to_gpu(x, device):
    return x.to(device, non_blocking=True)


_IndexOnly = rt.IndexOnly      # (name kept for the loader-order test)


def _join_process_group(local_rank, args):
    """-> (world, device).  One process per GPU over RCCL; the per-GPU batch is -b / #GPUs as in the reference (:168)."""
    if not (args.distributed and args.ngpus_per_node > 1):
        return 1, torch.device('cuda', local_rank if local_rank is not None else 0)
    rank = int(os.environ["RANK"]) if args.init_method == "env://" else local_rank
    dist.init_process_group(backend='nccl', init_method=args.init_method, world_size=args.ngpus_per_node, rank=rank)
    torch.cuda.set_device(local_rank)
    world = args.ngpus_per_node
    if args.batchSize < world:
        raise ValueError(f"-b {args.batchSize} with {world} GPUs gives a per-GPU batch of 0 (the reference needs -b >= #GPUs)")
    args.batchSize = args.batchSize // world
    return world, torch.device('cuda', local_rank)


def _make_loaders(args, world):
    """Full-clip loader (evaluation, unshuffled) and train loader (shuffled, drop_last) in the reference's construction order --
    the loaders consume the global RNG, and the frame order golden pins it.  With frames resident in HBM the loaders iterate
    indices only."""
    full = VideoDataSet(args)
    args.final_size, args.full_data_length = full.final_size, len(full)
    resident = not args.host_frames and not full.embed_inter
    items = rt.IndexOnly(len(full)) if resident else full
    workers = 0 if resident else args.workers
    sharded = torch.utils.data.distributed.DistributedSampler

    def loader(ds, batch, train):
        sampler = sharded(ds) if world > 1 else None
        return torch.utils.data.DataLoader(ds, batch_size=batch, shuffle=(train and sampler is None), num_workers=workers, pin_memory=True,
                                           sampler=sampler, drop_last=train, worker_init_fn=worker_init_fn)
    full_loader = loader(items, args.batchSize, False)
    train_ids, args.val_ind_list = data_split(list(range(args.full_data_length)), [int(x) for x in args.data_split.split('_')], args.shuffle_data, 0)
    train_loader = loader(Subset(items, train_ids), args.batchSize, True)
    return full, full_loader, train_loader, resident


def _make_optimizer(model, args):
    if args.optim_type == "Adam":
        return optim.Adam(model.parameters(), lr=args.lr)
    if args.optim_type == "Adan":
        from .optimizer import Adan
        return Adan(model.parameters(), lr=args.lr)
    raise ValueError(f"--optim_type {args.optim_type!r}: pass Adan or Adam (the reference's default 'adan' matches neither of its branches)")


def train(local_rank, args):
    for seed_fn in (torch.manual_seed, np.random.seed, random.seed):
        seed_fn(args.manualSeed)
    if not torch.cuda.is_available():
        raise RuntimeError("train_nerv_all: no ROCm GPU visible -- the decoder path has no CPU fallback")
    world, device = _join_process_group(local_rank, args)
    is_main = local_rank in (0, None)
    args.metric_names = list(rt.METRIC_NAMES)
    best = rt.BestTracker(args.metric_names)
    log = rt.RunLog(args.outf, is_main)

    full_dataset, full_loader, train_loader, resident = _make_loaders(args, world)
    args.dump_vis = args.dump_images or args.dump_videos
    args.fc_dim, embed_param = solve_fc_dim(args, args.final_size, args.full_data_length)
    model = build_model(args)

    writer = None
    if is_main:
        rt.dump_args(args, args.outf)
        args.encoder_param = sum(p.numel() for p in model.encoder.parameters()) / 1e6
        args.decoder_param = model.decoder_params()
        args.total_param = args.decoder_param + embed_param / 1e6
        sizes = f'Encoder_{round(args.encoder_param, 2)}M_Decoder_{round(args.decoder_param, 4)}M_Total_{round(args.total_param, 4)}M'
        print(f'{args}\n {sizes}', flush=True)
        log.line(f'{model}\n{sizes}', echo=False)
        try:
            from torch.utils.tensorboard import SummaryWriter
            writer = SummaryWriter(os.path.join(args.outf, sizes, 'tensorboard'))
        except Exception:
            writer = None

    print(f"Use GPU: {local_rank} for training")
    model = model.to(device)
    optimizer = _make_optimizer(model, args)
    args.transform_func = TransformInput(args)
    ckpt = rt.load_initial_state(model, args, args.outf, rename=lambda k: k.replace('blocks.0.', '').replace('module.', ''))
    if args.start_epoch < 0:
        args.start_epoch = max(ckpt['epoch'] if ckpt is not None else 0, 0)

    args._frames_dev = torch.stack([full_dataset[i]['img'] for i in range(len(full_dataset))]).to(device) if resident else None

    if args.eval_only:
        values, hw = evaluate(model, full_loader, local_rank, args, args.dump_vis, huffman_coding=True)
        top = best.update(values)
        text = f'PSNR for output {hw} for quant {args.quant_str}: ' + ''.join(
            f'best_{n}: {rt.fmt(v, 2 if "psnr" in n else 4)} | ' for n, v in zip(args.metric_names, top))
        if is_main:
            print(text, flush=True)
            with open(os.path.join(args.outf, 'eval.txt'), 'a') as f:
                f.write(text + '\n\n')
            args.train_time, args.cur_epoch = 0, args.epochs
            rt.write_results_csv(args, top, values, [torch.tensor(0)], 'eval.csv')
        return

    h, w = (int(x) for x in args.crop_list.split('_')[:2])
    takes_image = 'pe' not in args.embed or "HNeRV_Boost" in args.model
    fused = args.optim_type == "Adan" and args.transform_func.identity
    step = TrainStep(model, optimizer, args.loss, takes_image, (args.batchSize, 3, h, w), device, use_graph=not args.no_graph,
                     world_size=world, clip_max_norm=args.clip_max_norm) if fused else None
    # the generic path (Adam, inpainting masks) averages its gradients over the ranks with the same flat bucket the fused step uses
    bucket = GradBucket(model.parameters()) if (step is None and world > 1) else None

    t_start = time.time()
    epoch_secs, psnr_trace = [], []
    values = [torch.zeros(1) for _ in args.metric_names]
    lr = args.lr
    for epoch in range(args.start_epoch, args.epochs):
        model.train()
        t_epoch = time.time()
        psnr_sum = torch.zeros((), dtype=torch.float32, device=device)     # accumulated on the device: no per-step host sync
        seen = 0
        n_iter = len(train_loader)
        for i, sample in enumerate(train_loader):
            if args.debug and i > 10:
                break
            norm_idx, img_idx = data_to_gpu(sample['norm_idx'], device), data_to_gpu(sample['idx'], device)
            frames = args._frames_dev[img_idx] if resident else data_to_gpu(sample['img'], device)
            lr = adjust_lr(optimizer, (epoch + float(i) / n_iter) / args.epochs, i, args)
            if step is not None:
                _, psnr_b = step(frames, norm_idx)
            else:
                psnr_b = _generic_step(model, optimizer, bucket, args, frames, img_idx, norm_idx, takes_image)
            psnr_sum += psnr_b.sum()
            seen += psnr_b.numel()
            if i % args.print_freq == 0 or i == n_iter - 1:                 # the only host sync of the loop
                log.line('[{}] Rank:{}, Epoch[{}/{}], Step [{}/{}], lr:{:.2e} pred_PSNR: {}'.format(
                    log.stamp(), local_rank, epoch + 1, args.epochs, i + 1, n_iter, lr, rt.fmt((psnr_sum / seen).cpu(), 4)))
        train_psnr = psnr_sum / max(seen, 1)
        if world > 1:
            dist.all_reduce(train_psnr)
            train_psnr /= world
        if is_main:
            now = time.time()
            if writer is not None:
                writer.add_scalar(f'Train/pred_PSNR_{h}X{w}', train_psnr.cpu(), epoch + 1)
                writer.add_scalar('Train/lr', lr, epoch + 1)
            print("Time/epoch: \tCurrent:{:.2f} \tAverage:{:.2f}".format(now - t_epoch, (now - t_start) / (epoch + 1 - args.start_epoch)))
            epoch_secs.append(now - t_epoch)

        last = epoch == args.epochs - 1
        if (epoch + 1) % args.eval_freq == 0 or (args.epochs - epoch) in (1, 3, 5):
            values, hw = evaluate(model, full_loader, local_rank, args, args.dump_vis and last, last)
            if is_main:
                top = best.update(values)
                for name, v, b in zip(args.metric_names, values, top):
                    if 'psnr' in name and writer is not None:
                        writer.add_scalar(f'Val/{name}_{hw}', v.max(), epoch + 1)
                        writer.add_scalar(f'Val/best_{name}_{hw}', b, epoch + 1)
                psnr_trace.append(values[args.metric_names.index('pred_seen_psnr')].max())
                log.line(f'Eval at epoch {epoch + 1} for {hw}: ' + ''.join(f'{n}: {rt.fmt(v, 4)} | ' for n, v in zip(args.metric_names, values)))

        if is_main:
            state = {'epoch': epoch + 1, 'state_dict': model.state_dict(), 'optimizer': optimizer.state_dict()}
            torch.save(state, os.path.join(args.outf, 'model_latest.pth'))
            if last:
                args.cur_epoch, args.train_time = epoch + 1, rt.hms(time.time() - t_start)
                rt.write_results_csv(args, best.best, values, psnr_trace, f'epoch{epoch + 1}.csv')

    total = float(sum(epoch_secs))
    text = f"Training complete in: {rt.hms(time.time() - t_start)}\n Training wo evaluation complete in: {rt.hms(total)}, {total}s"
    print(text)
    if is_main:
        log.line(text, echo=False)
    if world > 1:
        dist.destroy_process_group()


def _generic_step(model, optimizer, bucket, args, frames, img_idx, norm_idx, takes_image):
    """One eager step for the configurations the captured step does not cover (--optim_type Adam, --inpanting masks): the same
    kernels, driven op by op.  With several ranks the gradients are averaged through the flat bucket before clipping / the step
    (the reference gets this from its DistributedDataParallel wrap)."""
    img_in, img_gt, mask = args.transform_func(frames, img_idx)
    out, _, _ = model(img_in if takes_image else norm_idx, norm_idx=norm_idx)
    loss = loss_fn(out, img_gt, args.loss) if mask is None else loss_fn(out * mask, img_gt * mask, args.loss)
    optimizer.zero_grad()
    loss.backward()
    if bucket is not None:
        bucket.allreduce_mean()
    if args.clip_max_norm > 0:
        torch.nn.utils.clip_grad_norm_(model.parameters(), args.clip_max_norm)
    optimizer.step()
    return psnr_fn_device(out.detach(), img_gt)


def _huffman_total_bits(counts):
    """sum_i count_i * code_length_i of a Huffman code built over the data symbols plus one EOF symbol of count 1 -- what
    the reference obtains from dahuffman's HuffmanCodec.from_data + get_code_table (train_nerv_all.py:593-605).  Every optimal
    code has the same total over all leaves; which tie the EOF leaf wins can move the data-only total by a bit or two."""
    n = len(counts)
    heap = [(int(c), i, [i]) for i, c in enumerate(list(counts) + [1])]      # last leaf = EOF
    depth = [0] * (n + 1)
    heapq.heapify(heap)
    uid = n + 1
    while len(heap) > 1:
        a = heapq.heappop(heap)
        b = heapq.heappop(heap)
        for leaf in a[2] + b[2]:
            depth[leaf] += 1
        heapq.heappush(heap, (a[0] + b[0], uid, a[2] + b[2]))
        uid += 1
    return sum(int(c) * d for c, d in zip(counts, depth[:n]))


def quant_model(model, args):
    """[fp32 copy, post-hoc quantised twin], {key: quantised tensor record} -- the two models evaluate() reports on
    (reference train_nerv_all.py:622-642).  Every decoder tensor goes through hnerv_utils.quant_tensor at --quant_model_bit bits;
    the encoder is left alone (it is not part of the transmitted model).  --quant_model_bit -1 disables the twin."""
    plain = copy.deepcopy(model)
    if args.quant_model_bit == -1:
        return [plain], None
    twin = copy.deepcopy(model)
    records, dequantised = {}, {}
    for key, tensor in twin.state_dict().items():
        if 'encoder' in key:
            dequantised[key] = tensor
        else:
            records[key], dequantised[key] = quant_tensor(tensor, args.quant_model_bit)
    twin.load_state_dict(dequantised)
    return [plain, twin], records


def _timed_decodes(net, cur_input, embed, norm_idx, graph):
    """100 repeated decodes of one batch for --eval_fps; a captured hipGraph of the same forward unless BNERV_EVAL_GRAPH=0."""
    times = []
    if cur_input.is_cuda and os.environ.get('BNERV_EVAL_GRAPH', '1') != '0':
        from .engine import DecodeGraph
        if graph is None or not graph.matches(cur_input, embed, norm_idx):
            graph = DecodeGraph(net, cur_input, embed, norm_idx)
        for _ in range(100):
            times.append(graph(cur_input, embed, norm_idx)[1])
    else:
        for _ in range(100):
            times.append(net(cur_input, embed, norm_idx=norm_idx)[2])
    return times, graph


@torch.no_grad()
def evaluate(model, full_dataloader, local_rank, args, dump_vis=False, huffman_coding=False):
    """PSNR / MS-SSIM of the fp32 model and of its post-hoc quantised twin over the whole clip (seen / unseen frames apart),
    decode FPS, and -- on request -- the Huffman bit accounting of the quantised tensors.  Returns (values per metric slot, (h, w))."""
    nets, records = quant_model(model, args)
    book = rt.MetricBook(args.val_ind_list, args.metric_names)
    log = rt.RunLog(args.outf, local_rank in (0, None))
    frames_dev = getattr(args, '_frames_dev', None)
    takes_image = 'pe' not in args.embed or "HNeRV_Boost" in args.model
    is_hnerv = "HNeRV" in args.model
    embeds, dequant_embeds, quant_embed = [], None, None
    fps, hw = 0.0, (0, 0)
    for slot, net in enumerate(nets):
        net.eval()
        net.time_decode = True
        device = next(net.parameters()).device
        times, graph = [], None
        vis_dir = None
        if dump_vis:
            vis_dir = os.path.join(args.outf, 'visualize_model' + ('_quant' if slot else '_orig'))
            os.makedirs(vis_dir, exist_ok=True)
        n_batches = len(full_dataloader)
        for i, sample in enumerate(full_dataloader):
            if args.debug and i > 10:
                break
            norm_idx, img_idx = data_to_gpu(sample['norm_idx'], device), data_to_gpu(sample['idx'], device)
            frames = frames_dev[img_idx] if frames_dev is not None else data_to_gpu(sample['img'], device)
            img_in, img_gt, _mask = args.transform_func(frames, img_idx)
            cur_input = img_in if takes_image else norm_idx
            embed_in = dequant_embeds[i] if (slot and is_hnerv) else None
            extra = {}
            if args.interpolation and args.embed_inter and 'pre_img' in sample and img_idx.item() in args.val_ind_list:
                extra = dict(pre_img=data_to_gpu(sample['pre_img'], device), post_img=data_to_gpu(sample['post_img'], device))
            out, embed_list, dec_time = net(cur_input, embed_in, norm_idx=norm_idx, **extra)
            if slot == 0:
                embeds.append(embed_list[0])
            if args.eval_fps:
                more, graph = _timed_decodes(net, cur_input, embed_list[0], norm_idx, graph)
                times.extend(more)
            else:
                times.append(dec_time)
            psnr, ssim = ops.psnr(out, img_gt), ops.msssim(out.float(), img_gt)     # stay on the device until a line is printed
            book.add(slot, sample['idx'].tolist(), psnr, ssim)
            if dump_vis:
                _save_images(out, psnr, vis_dir, i, args)
            if i % args.print_freq == 0 or i == n_batches - 1:
                fps = args.batchSize / (sum(times) / len(times))
                log.line(f'[{log.stamp()}] Rank:{local_rank}, Eval at Step [{i + 1}/{n_batches}] , FPS {round(fps, 1)}, ' + book.describe(book.running()),
                         every_rank=False)
        if slot == 0 and is_hnerv:
            quant_embed, deq = quant_tensor(torch.cat(embeds, 0), args.quant_embed_bit)
            dequant_embeds = deq.split(args.batchSize, dim=0)
        args.fps, hw = fps, tuple(frames.shape[-2:])
        net.time_decode = False
        net.train()
    values = book.means(device=next(model.parameters()).device)
    if local_rank in (0, None) and records is not None and huffman_coding:
        _huffman_report(args, records, quant_embed if is_hnerv else None, log)
    return values, hw


def _huffman_report(args, records, quant_embed, log):
    """Bits per parameter / per pixel of the post-hoc quantised tensors under one Huffman code over all symbols, plus 16 bits for
    every stored (min, scale) entry (reference train_nerv_all.py:577-612)."""
    symbols, side_entries = [], 0
    for rec in ([quant_embed] if quant_embed is not None else []) + list(records.values()):
        symbols.append(rec['quant'].flatten().cpu())
        side_entries += rec['min'].nelement() + rec['scale'].nelement()
    flat = torch.cat(symbols).to(torch.int64)
    hist = torch.bincount(flat, minlength=1)
    bits = _huffman_total_bits(hist[hist > 0].tolist())
    args.bits_per_param = bits / flat.numel()
    bits += side_entries * 16
    args.full_bits_per_param = bits / flat.numel()
    args.total_bpp = bits / args.final_size / args.full_data_length
    log.line(f'After quantization and encoding: \n bits per parameter: {round(args.full_bits_per_param, 2)}, bits per pixel: {round(args.total_bpp, 4)}')


def _save_images(out, psnr, vis_dir, batch_no, args):
    from PIL import Image
    for b in range(out.shape[0]):
        arr = (out[b].clamp(0, 1) * 255 + 0.5).to(torch.uint8).permute(1, 2, 0).cpu().numpy()
        Image.fromarray(arr).save(os.path.join(vis_dir, f'pred_{batch_no * args.batchSize + b:04d}_{round(psnr[b].item(), 2)}.png'))


Dump2CSV = rt.write_results_csv      # (the reference's name for the results table writer)


if __name__ == '__main__':
    main()
