"""Compression (CEM) training and evaluation -- the reference's train_nerv_compression.py CLI on the MI355X build (SURVEY 8(f)
row N2, BASELINE configs[4]).  Same flags (train_nerv_compression.py:28-116), same per-step sequence (:340-367):

    adjust_lr -> model.cal_params(entropy_model) -> forward (embedding rate when --embed_entropy) -> loss_fn(...) [+ lambda * bpp
    while bpp / N > target_bpp] -> backward -> optimizer.step

and the same evaluation report (:457-585): PSNR / MS-SSIM of the model run with its de-quantised weights and embeddings,
estimated and "real" bits per pixel.  The decoder runs on the HIP kernels; the quantise / rate arithmetic on the weights is
tensor-at-a-time device code (lib/transform_ops.py, lib/entropy_model.py).  Real bits: ideal code length (the reference's ANS
coder, constriction, is not available here -- see lib/entropy_model.py)."""
import os
import random
import shutil
import time

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from . import ops
from . import runtime as rt
from . import train_nerv_all as T
from .dp import GradBucket
from .hnerv_utils import TransformInput, adjust_lr, loss_fn
from .lib.entropy_model import DiffEntropyModel
from .lib.quant_ops import CustomConv2d, CustomLinear


def build_parser():
    p = T.build_parser()
    have = {a.dest for a in p._actions}
    def add(flag, **kw):
        if flag.lstrip('-') not in have:
            p.add_argument(flag, **kw)
    add('--quant', action='store_true', default=False, help='enable quantization')
    add('--quant_bias_bit', type=int, default=8, help='bit length for bias quantization')
    add('--per_channel_w', action='store_true', default=False)
    add('--per_channel_b', action='store_true', default=False)
    add('--per_channel_e', action='store_true', default=False)
    add('--quantizer_w', type=str, default='lsq')
    add('--quantizer_b', type=str, default='lsq')
    add('--quantizer_e', type=str, default='lsqv2')
    add('--embed_entropy', action='store_true', default=False, help='use entropy model for embedding')
    add('--target_bit', type=float, default=5)
    add('--lambda_rate', type=float, default=0.2)
    return p


def data_to_gpu(x, device):
    return x.to(device, non_blocking=True)


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
    if args.debug:      # the reference's debug runs spell the architecture out in the directory name
        args.exp_id += 'Size{}_ENC_{}_{}_DEC_{}_{}_{}{}{}'.format(args.modelsize, args.conv_type[0], args.enc_strd_str, args.conv_type[1], args.dec_strd_str,
                                                                 '' if args.norm == 'none' else f'_{args.norm}', '_dist' if args.distributed else '',
                                                                 '_shuffle_data' if args.shuffle_data else '')
    args.outf = os.path.join(args.outf, args.exp_id)
    if args.overwrite and os.path.isdir(args.outf):
        shutil.rmtree(args.outf)
    os.makedirs(args.outf, exist_ok=True)
    args.ngpus_per_node = torch.cuda.device_count()
    env_world = int(os.environ.get("WORLD_SIZE", "1"))
    if env_world > 1:                                      # torchrun: one process per GPU exists already
        args.distributed, args.ngpus_per_node, args.init_method = True, env_world, "env://"
        train(int(os.environ.get("LOCAL_RANK", 0)), args)
    elif args.distributed and args.ngpus_per_node > 1:
        args.init_method = f'tcp://127.0.0.1:{hash(args.exp_id) % 20000 + 10000}'
        mp.spawn(train, nprocs=args.ngpus_per_node, args=(args,))
    else:
        train(None, args)


def _quant_modules(model):
    return [m for m in model.modules() if type(m) in (CustomConv2d, CustomLinear)]


def _rate_terms(model, entropy_model, args, cur_input, norm_idx):
    """forward + bits per pixel (x N) of the model and, with --embed_entropy, of this batch's embedding scaled to the clip."""
    if args.embed_entropy:
        out, _, _ = model(cur_input, entropy_model=entropy_model, norm_idx=norm_idx)
        bits = model.get_bitrate_sum(name="bitrate") + model.bitrate_e_dict["bitrate"] * args.full_data_length
    else:
        out, _, _ = model(cur_input, norm_idx=norm_idx)
        bits = model.get_bitrate_sum(name="bitrate")
    return out, bits / args.final_size


def _eager_rd_step(model, optimizer, entropy_model, bucket, args, frames, img_idx, norm_idx, takes_image):
    """The rate-distortion step driven op by op (inpainting masks, Adam, --no_graph): quantise + rate of every tensor, forward,
    distortion + lambda * bpp while the rate is above the target (decided on the device), backward, [gradient mean over the
    ranks], optimizer."""
    img_in, img_gt, mask = args.transform_func(frames, img_idx)
    model.cal_params(entropy_model)
    out, bpp = _rate_terms(model, entropy_model, args, img_in if takes_image else norm_idx, norm_idx)
    dist_loss = loss_fn(out, img_gt, args.loss) if mask is None else loss_fn(out * mask, img_gt * mask, args.loss)
    over_budget = (bpp.detach() / args.full_data_length > args.target_bpp).to(dist_loss.dtype)
    loss = dist_loss + over_budget * args.lambda_rate * bpp
    optimizer.zero_grad()
    loss.backward()
    if bucket is not None:
        bucket.allreduce_mean()
    if args.clip_max_norm > 0:
        torch.nn.utils.clip_grad_norm_(model.parameters(), args.clip_max_norm)
    optimizer.step()
    return loss.detach(), ops.psnr(out.detach(), img_gt), bpp.detach()


def train(local_rank, args):
    for seed_fn in (torch.manual_seed, np.random.seed, random.seed):
        seed_fn(args.manualSeed)
    if not torch.cuda.is_available():
        raise RuntimeError("train_nerv_compression: no ROCm GPU visible -- the decoder path has no CPU fallback")
    world, device = T._join_process_group(local_rank, args)
    is_main = local_rank in (0, None)
    args.metric_names = list(rt.METRIC_NAMES)
    best = rt.BestTracker(args.metric_names)
    log = rt.RunLog(args.outf, is_main)

    full_dataset, full_loader, train_loader, resident = T._make_loaders(args, world)
    args.dump_vis = args.dump_images or args.dump_videos
    args.fc_dim, embed_param = T.solve_fc_dim(args, args.final_size, args.full_data_length)
    model = T.build_model(args)
    entropy_model = DiffEntropyModel(distribution="gaussian")

    args.encoder_param = sum(p.numel() for p in model.encoder.parameters()) / 1e6
    args.decoder_param = model.decoder_params()
    args.total_param = args.decoder_param + embed_param / 1e6
    # the rate budget: --target_bit bits per parameter spread over the pixels of the clip (reference train_nerv_compression.py:252)
    args.target_bpp = args.target_bit * args.total_param * 1e6 / args.final_size / args.full_data_length
    if is_main:
        rt.dump_args(args, args.outf)
        sizes = f'Encoder_{round(args.encoder_param, 2)}M_Decoder_{round(args.decoder_param, 4)}M_Total_{round(args.total_param, 4)}M'
        print(f'{args}\n {sizes}', flush=True)
        log.line(f'{model}\n{sizes}', echo=False)

    model = model.to(device)
    optimizer = T._make_optimizer(model, args)
    args.transform_func = TransformInput(args)
    # a regression checkpoint (--weight) has no quantiser parameters, a resumed compression checkpoint has them: both load non-strictly
    ckpt = rt.load_initial_state(model, args, args.outf, strict_resume=False, rename=lambda k: k.replace('module.', ''))
    if args.start_epoch < 0:
        args.start_epoch = max(ckpt['epoch'] if (ckpt is not None and not args.not_resume) else 0, 0)

    args._frames_dev = torch.stack([full_dataset[i]['img'] for i in range(len(full_dataset))]).to(device) if resident else None
    takes_image = 'pe' not in args.embed or "HNeRV_Boost" in args.model

    if args.eval_only:
        # (no init_data() here: the quantiser scales come from the checkpoint -- re-initialising them from the weight range would
        #  report the rate / quality of a different model; the reference initialises them on the training path only)
        values, hw = evaluate(model, full_loader, local_rank, args, args.dump_vis, coding=True, entropy_model=entropy_model)
        top = rt.BestTracker(args.metric_names).update(values)
        text = f'PSNR for output {hw} for quant {args.quant_str}: ' + ''.join(
            f'best_{n}: {rt.fmt(v, 4)} | ' for n, v in zip(args.metric_names, top))      # 4 decimals for every metric (train_nerv_compression.py:317)
        if is_main:
            print(text, flush=True)
            # the files the reference's --eval_only writes under --outf (train_nerv_compression.py:311-325): eval.txt and eval.csv
            with open(os.path.join(args.outf, 'eval.txt'), 'a') as f:
                f.write(text + '\n\n')
            args.train_time, args.cur_epoch = 0, args.epochs
            rt.write_results_csv(args, top, values, [torch.tensor(0)], 'eval.csv')
        return

    model.init_data()                                                  # quantiser scales from the current weight ranges (:338)
    h_, w_ = (int(v) for v in args.crop_list.split('_')[:2])
    fused = hasattr(optimizer, 'prepare_step') and args.transform_func.identity and args.clip_max_norm <= 0 and not getattr(args, 'no_graph', False)
    cstep = None
    if fused:
        from .engine import CompressionStep
        cstep = CompressionStep(model, optimizer, entropy_model, args, (args.batchSize, 3, h_, w_), device, world_size=world)
    bucket = GradBucket(model.parameters()) if (cstep is None and world > 1) else None

    t_start = time.time()
    epoch_secs, psnr_trace = [], []
    values = [torch.zeros(1) for _ in args.metric_names]
    for epoch in range(args.start_epoch, args.epochs):
        model.train()
        t_epoch = time.time()
        psnr_sum, seen = torch.zeros((), dtype=torch.float32, device=device), 0
        n_iter = len(train_loader)
        for i, sample in enumerate(train_loader):
            if args.debug and i > 10:
                break
            norm_idx, img_idx = data_to_gpu(sample['norm_idx'], device), data_to_gpu(sample['idx'], device)
            frames = args._frames_dev[img_idx] if resident else data_to_gpu(sample['img'], device)
            lr = adjust_lr(optimizer, (epoch + float(i) / n_iter) / args.epochs, i, args)
            if cstep is not None:          # the whole step as one captured graph (engine.CompressionStep)
                loss, psnr_b = cstep(frames, norm_idx)
                bpp = cstep.bpp_out
            else:
                loss, psnr_b, bpp = _eager_rd_step(model, optimizer, entropy_model, bucket, args, frames, img_idx, norm_idx, takes_image)
            psnr_sum += psnr_b.sum()
            seen += psnr_b.numel()
            if i % args.print_freq == 0 or i == n_iter - 1:
                log.line('[{}] Rank:{}, Epoch[{}/{}], Step [{}/{}], lr:{:.2e} pred_PSNR: {}, loss:{}, bpp:{}'.format(
                    log.stamp(), local_rank, epoch + 1, args.epochs, i + 1, n_iter, lr, rt.fmt((psnr_sum / seen).cpu(), 2),
                    rt.fmt(loss.cpu(), 4), rt.fmt((bpp / args.full_data_length).cpu(), 6)))
        now = time.time()
        if is_main:
            print("Time/epoch: \tCurrent:{:.2f} \tAverage:{:.2f}".format(now - t_epoch, (now - t_start) / (epoch + 1 - args.start_epoch)))
        epoch_secs.append(now - t_epoch)
        last = epoch == args.epochs - 1
        if (epoch + 1) % args.eval_freq == 0 or (args.epochs - epoch) in (1, 3, 5):
            values, hw = evaluate(model, full_loader, local_rank, args, args.dump_vis and last, coding=True, entropy_model=entropy_model)
            best.update(values)
            psnr_trace.append(values[args.metric_names.index('quant_seen_psnr')].max())
            if is_main:
                log.line(f'Eval at epoch {epoch + 1} for {hw}: ' + ''.join(f'{n}: {rt.fmt(v, 4)} | ' for n, v in zip(args.metric_names, values)))
        if is_main:
            state = {'epoch': epoch + 1, 'state_dict': model.state_dict(), 'optimizer': optimizer.state_dict()}
            torch.save(state, os.path.join(args.outf, 'model_latest.pth'))
            if last:       # the artefacts downstream evaluation scripts pass through --weight (reference :431-433)
                torch.save(state, os.path.join(args.outf, f'epoch{epoch + 1}.pth'))
                torch.save(state, os.path.join(args.outf, 'model_best.pth'))
                args.cur_epoch, args.train_time = epoch + 1, rt.hms(time.time() - t_start)
                rt.write_results_csv(args, best.best, values, psnr_trace, f'epoch{epoch + 1}.csv')
    print(f"Training complete in: {rt.hms(time.time() - t_start)}\n Training wo evaluation complete in: {float(sum(epoch_secs))}s")
    if world > 1:
        dist.destroy_process_group()


@torch.no_grad()
def evaluate(model, full_dataloader, local_rank, args, dump_vis=False, coding=False, entropy_model=None):
    """Quality of the model run with its DE-QUANTISED weights and embeddings (filed under the quant_* metric slots, as the
    reference does, train_nerv_compression.py:457-585) and the bit accounting: bits the rate model estimates for the rounded
    symbols, and the bytes an ANS coder actually produces for them (lib/entropy_model.py -> csrc/ans.cpp), plus 32 bits for every
    transmitted quantiser / entropy-model parameter."""
    book = rt.MetricBook(args.val_ind_list, args.metric_names)
    log = rt.RunLog(args.outf, local_rank in (0, None))
    model.eval()
    model.time_decode = True
    model._fused_bits_total = None          # the rate dictionaries are refilled below: a cached training-step total must not be reported
    device = next(model.parameters()).device
    side_params, n_stats = 0, 0
    # every tensor's rounded symbols, estimated bits and coded bits in one fused pass + one host copy (model_nerv._CEMHooks); the
    # tensor-by-tensor loop of the reference (:466-489) below serves the settings the fused kernel does not (BNERV_CEM_EVAL_FUSED=0 forces it)
    fused = (os.environ.get("BNERV_CEM_EVAL_FUSED", "1") != "0" and getattr(model, "cem_fused", True)
             and hasattr(model, "cal_params_eval_fused") and model.cal_params_eval_fused(entropy_model))
    for m in _quant_modules(model):
        for kind in ('weight', 'bias'):
            tensor = getattr(m, kind)
            if tensor is None:
                continue
            quantizer = getattr(m, f'{kind}_quantizer')
            side_params += sum(p.numel() for p in quantizer.parameters())
            if entropy_model is not None:
                n_stats += 2                # (mean, std) of the tensor's Gaussian
            if fused:
                continue
            code, symbols, dequant = quantizer(tensor)
            setattr(m, 'dequant_w' if kind == 'weight' else 'dequant_b', dequant)
            if entropy_model is not None:
                getattr(m, 'bitrate_w_dict' if kind == 'weight' else 'bitrate_b_dict').update(entropy_model.cal_bitrate(code, symbols, False))
    is_hnerv = "HNeRV" in args.model
    takes_image = 'pe' not in args.embed or "HNeRV_Boost" in args.model
    if is_hnerv:
        side_params += sum(p.numel() for p in model.embed_quantizer.parameters())
    embed_est, embed_real, embed_stats = 0.0, 0.0, 0
    frames_dev = getattr(args, '_frames_dev', None)
    times, fps, frames = [], 0.0, None
    n_batches = len(full_dataloader)
    for i, sample in enumerate(full_dataloader):
        norm_idx, img_idx = data_to_gpu(sample['norm_idx'], device), data_to_gpu(sample['idx'], device)
        frames = frames_dev[img_idx] if frames_dev is not None else data_to_gpu(sample['img'], device)
        img_in, img_gt, _mask = args.transform_func(frames, img_idx)
        if is_hnerv:
            code_e, symbols_e, dequant_e = model.forward_embed_quant(model.forward_encoder(img_in))
            if args.embed_entropy:
                r = entropy_model.cal_bitrate(code_e, symbols_e, False)
                embed_est += float(r["bitrate"])
                embed_real += r["real_bitrate"]
                embed_stats += 2
            out, _, dec_time = model.forward_decoder(dequant_e, norm_idx)
        else:
            out, _, dec_time = model(img_in if takes_image else norm_idx, norm_idx=norm_idx)
        times.append(dec_time)
        book.add(1, sample['idx'].tolist(), ops.psnr(out, img_gt), ops.msssim(out.float(), img_gt))
        if i % args.print_freq == 0 or i == n_batches - 1:
            fps = args.batchSize / (sum(times) / len(times))
            log.line(f'[{log.stamp()}] Rank:{local_rank}, Eval at Step [{i + 1}/{n_batches}] , FPS {round(fps, 2)}, ' + book.describe(book.running()),
                     every_rank=False)
    values = book.means(device=device)
    args.fps, hw = fps, tuple(frames.shape[-2:])
    model.time_decode = False
    model.train()
    if coding:
        pixels = args.final_size * args.full_data_length
        est_bits = float(model.get_bitrate_sum(name="bitrate")) + (embed_est if is_hnerv else 0.0)
        coded_bits = float(model.get_bitrate_sum(name="real_bitrate")) + (embed_real if is_hnerv else 0.0)
        side_bits = (n_stats + side_params + (embed_stats if is_hnerv else 0)) * 32
        args.total_bpp = (coded_bits + side_bits) / pixels
        args.estimate_bpp = (est_bits + side_bits) / pixels
        log.line(f'Gaussian Entropy Model real bpp: {round(args.total_bpp, 6)}, estimated bpp:{round(args.estimate_bpp, 6)}, '
                 f'target_bpp:{round(args.target_bpp, 6)} \n', every_rank=False)
    return values, hw


if __name__ == '__main__':
    main()
