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
from datetime import datetime

import numpy as np
import torch
import torch.optim as optim
import yaml
from torch.utils.data import Subset

from . import ops
from . import train_nerv_all as T
from .hnerv_utils import RoundTensor, TransformInput, VideoDataSet, adjust_lr, data_split, loss_fn, worker_init_fn
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
    torch.set_printoptions(precision=4)
    if args.debug:
        args.eval_freq = 1
        args.outf = 'output/debug'
    else:
        args.outf = os.path.join('output', args.outf)
    args.enc_strd_str, args.dec_strd_str = ','.join([str(x) for x in args.enc_strds]), ','.join([str(x) for x in args.dec_strds])
    extra_str = 'Size{}_ENC_{}_{}_DEC_{}_{}_{}{}{}'.format(args.modelsize, args.conv_type[0], args.enc_strd_str, args.conv_type[1], args.dec_strd_str,
                                                           '' if args.norm == 'none' else f'_{args.norm}', '_dist' if args.distributed else '',
                                                           '_shuffle_data' if args.shuffle_data else '')
    args.quant_str = f'quant_M{args.quant_model_bit}_E{args.quant_embed_bit}'
    args.exp_id = f'{args.vid}/Size{args.modelsize}'
    args.exp_id += extra_str if args.debug else ''
    args.outf = os.path.join(args.outf, args.exp_id)
    if args.overwrite and os.path.isdir(args.outf):
        import shutil
        shutil.rmtree(args.outf)
    os.makedirs(args.outf, exist_ok=True)
    torch.set_printoptions(precision=2)
    args.ngpus_per_node = torch.cuda.device_count()
    if args.distributed and args.ngpus_per_node > 1:
        raise NotImplementedError("train_nerv_compression: the multi-GPU launch is only wired for train_nerv_all.py in this build")
    train(None, args)


def _quant_modules(model):
    return [m for m in model.modules() if type(m) in (CustomConv2d, CustomLinear)]


def train(local_rank, args):
    torch.manual_seed(args.manualSeed)
    np.random.seed(args.manualSeed)
    random.seed(args.manualSeed)
    if not torch.cuda.is_available():
        raise RuntimeError("train_nerv_compression: no ROCm GPU visible -- the decoder path has no CPU fallback")
    torch.backends.cudnn.benchmark = True
    device = torch.device('cuda', 0)
    args.metric_names = ['pred_seen_psnr', 'pred_seen_ssim', 'pred_unseen_psnr', 'pred_unseen_ssim',
                         'quant_seen_psnr', 'quant_seen_ssim', 'quant_unseen_psnr', 'quant_unseen_ssim']
    best_metric_list = [torch.tensor(0) for _ in range(len(args.metric_names))]

    full_dataset = VideoDataSet(args)
    args.final_size = full_dataset.final_size
    args.full_data_length = len(full_dataset)
    resident = not args.host_frames and not full_dataset.embed_inter
    loader_ds = T._IndexOnly(len(full_dataset)) if resident else full_dataset
    workers = 0 if resident else args.workers
    full_dataloader = torch.utils.data.DataLoader(loader_ds, batch_size=args.batchSize, shuffle=False, num_workers=workers, pin_memory=True,
                                                  drop_last=False, worker_init_fn=worker_init_fn)
    split_num_list = [int(x) for x in args.data_split.split('_')]
    train_ind_list, args.val_ind_list = data_split(list(range(args.full_data_length)), split_num_list, args.shuffle_data, 0)
    args.dump_vis = (args.dump_images or args.dump_videos)
    train_dataloader = torch.utils.data.DataLoader(Subset(loader_ds, train_ind_list), batch_size=args.batchSize, shuffle=True, num_workers=workers,
                                                   pin_memory=True, drop_last=True, worker_init_fn=worker_init_fn)

    args.fc_dim, embed_param = T.solve_fc_dim(args, args.final_size, args.full_data_length)
    model = T.build_model(args)
    entropy_model = DiffEntropyModel(distribution="gaussian")

    with open(os.path.join(args.outf, 'args.yaml'), 'w') as f:
        f.write(yaml.safe_dump({k: v for k, v in args.__dict__.items() if isinstance(v, (int, float, str, bool, list, type(None)))},
                               default_flow_style=False))
    encoder_param = (sum([p.data.nelement() for p in model.encoder.parameters()]) / 1e6)
    decoder_param = model.decoder_params()
    total_param = decoder_param + embed_param / 1e6
    args.encoder_param, args.decoder_param, args.total_param = encoder_param, decoder_param, total_param
    args.target_bpp = args.target_bit * args.total_param * 1e6 / args.final_size / args.full_data_length     # train_nerv_compression.py:252
    param_str = f'Encoder_{round(encoder_param, 2)}M_Decoder_{round(decoder_param, 4)}M_Total_{round(total_param, 4)}M'
    print(f'{args}\n {param_str}', flush=True)
    with open('{}/rank0.txt'.format(args.outf), 'a') as f:
        f.write(str(model) + '\n' + f'{param_str}\n')

    model = model.to(device)
    if args.optim_type == "Adam":
        optimizer = optim.Adam(model.parameters(), lr=args.lr)
    elif args.optim_type == "Adan":
        from .optimizer import Adan
        optimizer = Adan(model.parameters(), lr=args.lr)
    else:
        raise ValueError(f"--optim_type {args.optim_type!r}: pass Adan or Adam")
    args.transform_func = TransformInput(args)

    checkpoint = None
    if args.weight != 'None':
        print("=> loading checkpoint '{}'".format(args.weight))
        checkpoint = torch.load(args.weight, map_location='cpu')
        ckt = {k.replace('module.', ''): v for k, v in checkpoint['state_dict'].items()}
        model.load_state_dict(ckt, strict=False)                   # the regression checkpoint has no quantiser parameters (:288-296)
        print("=> loaded checkpoint '{}' (epoch {})".format(args.weight, checkpoint['epoch']))
    if not args.not_resume:
        checkpoint_path = os.path.join(args.outf, 'model_latest.pth')
        if os.path.isfile(checkpoint_path):
            checkpoint = torch.load(checkpoint_path, map_location='cpu')
            model.load_state_dict(checkpoint['state_dict'], strict=False)
            print("=> Auto resume loaded checkpoint '{}' (epoch {})".format(checkpoint_path, checkpoint['epoch']))
        else:
            print("=> No resume checkpoint found at '{}'".format(checkpoint_path))
    if args.start_epoch < 0:
        if checkpoint is not None and not args.not_resume:
            args.start_epoch = checkpoint['epoch']
        args.start_epoch = max(args.start_epoch, 0)

    frames_dev = torch.stack([full_dataset[i]['img'] for i in range(len(full_dataset))]).to(device) if resident else None
    args._frames_dev = frames_dev
    takes_image = 'pe' not in args.embed or "HNeRV_Boost" in args.model

    if args.eval_only:
        model.init_data()
        results_list, hw = evaluate(model, full_dataloader, local_rank, args, args.dump_vis, coding=True, entropy_model=entropy_model)
        print(f'PSNR for output {hw} for quant {args.quant_str}: ' + ' | '.join(f'{n}: {RoundTensor(v, 4)}' for n, v in zip(args.metric_names, results_list)), flush=True)
        return

    start = datetime.now()
    psnr_list, time_list = [], []
    results_list = [torch.zeros(1) for _ in args.metric_names]
    model.init_data()                                                  # :338
    cstep = None
    if device.type == 'cuda' and not getattr(args, 'no_graph', False) and args.clip_max_norm <= 0 and hasattr(optimizer, 'prepare_step'):
        from .engine import CompressionStep
        h_, w_ = [int(v) for v in args.crop_list.split('_')[:2]]
        cstep = CompressionStep(model, optimizer, entropy_model, args, (args.batchSize, 3, h_, w_), device)
    for epoch in range(args.start_epoch, args.epochs):
        model.train()
        epoch_start_time = datetime.now()
        psnr_sum, psnr_cnt = torch.zeros((), dtype=torch.float32, device=device), 0
        n_iter = len(train_dataloader)
        for i, sample in enumerate(train_dataloader):
            if i > 10 and args.debug:
                break
            norm_idx, img_idx = data_to_gpu(sample['norm_idx'], device), data_to_gpu(sample['idx'], device)
            img_data = frames_dev[img_idx] if resident else data_to_gpu(sample['img'], device)
            img_data, img_gt, inpaint_mask = args.transform_func(img_data, img_idx)
            cur_input = img_data if takes_image else norm_idx
            cur_epoch = (epoch + float(i) / n_iter) / args.epochs
            lr = adjust_lr(optimizer, cur_epoch, i, args)
            if cstep is not None and inpaint_mask is None and img_gt is img_data:
                # the whole step as one captured graph (engine.CompressionStep; --no_graph keeps the eager sequence below)
                final_loss, psnr_b = cstep(img_data, norm_idx)
                bpp = cstep.bpp_out
                psnr_sum += psnr_b.sum()
                psnr_cnt += psnr_b.numel()
                if i % args.print_freq == 0 or i == n_iter - 1:
                    print_str = '[{}] Rank:{}, Epoch[{}/{}], Step [{}/{}], lr:{:.2e} pred_PSNR: {}, loss:{}, bpp:{}'.format(
                        datetime.now().strftime("%Y/%m/%d %H:%M:%S"), local_rank, epoch + 1, args.epochs, i + 1, n_iter, lr,
                        RoundTensor((psnr_sum / psnr_cnt).cpu(), 2), RoundTensor(final_loss.detach().cpu(), 4),
                        RoundTensor((bpp.detach() / args.full_data_length).cpu(), 6))
                    print(print_str, flush=True)
                    with open('{}/rank0.txt'.format(args.outf), 'a') as f:
                        f.write(print_str + '\n')
                continue
            model.cal_params(entropy_model)
            if args.embed_entropy:
                img_out, _, _ = model(cur_input, entropy_model=entropy_model, norm_idx=norm_idx)
                bit_embed = model.bitrate_e_dict["bitrate"] * args.full_data_length
                bpp = (model.get_bitrate_sum(name="bitrate") + bit_embed) / args.final_size
            else:
                img_out, _, _ = model(cur_input, norm_idx=norm_idx)
                bpp = model.get_bitrate_sum(name="bitrate") / args.final_size
            out_loss = loss_fn(img_out, img_gt, args.loss) if inpaint_mask is None else loss_fn(img_out * inpaint_mask, img_gt * inpaint_mask, args.loss)
            # `if bpp / N > target_bpp` (:363) decided on the device: no host sync inside the step
            gate = (bpp.detach() / args.full_data_length > args.target_bpp).to(out_loss.dtype)
            final_loss = out_loss + gate * args.lambda_rate * bpp
            optimizer.zero_grad()
            final_loss.backward()
            if args.clip_max_norm > 0:
                torch.nn.utils.clip_grad_norm_(model.parameters(), args.clip_max_norm)
            optimizer.step()
            psnr_b = ops.psnr(img_out.detach(), img_gt)
            psnr_sum += psnr_b.sum()
            psnr_cnt += psnr_b.numel()
            if i % args.print_freq == 0 or i == n_iter - 1:
                print_str = '[{}] Rank:{}, Epoch[{}/{}], Step [{}/{}], lr:{:.2e} pred_PSNR: {}, loss:{}, bpp:{}'.format(
                    datetime.now().strftime("%Y/%m/%d %H:%M:%S"), local_rank, epoch + 1, args.epochs, i + 1, n_iter, lr,
                    RoundTensor((psnr_sum / psnr_cnt).cpu(), 2), RoundTensor(final_loss.detach().cpu(), 4),
                    RoundTensor((bpp.detach() / args.full_data_length).cpu(), 6))
                print(print_str, flush=True)
                with open('{}/rank0.txt'.format(args.outf), 'a') as f:
                    f.write(print_str + '\n')
        epoch_end_time = datetime.now()
        print("Time/epoch: \tCurrent:{:.2f} \tAverage:{:.2f}".format((epoch_end_time - epoch_start_time).total_seconds(),
                                                                    (epoch_end_time - start).total_seconds() / (epoch + 1 - args.start_epoch)))
        time_list.append((epoch_end_time - epoch_start_time).total_seconds())
        if (epoch + 1) % args.eval_freq == 0 or (args.epochs - epoch) in [1, 3, 5]:
            results_list, hw = evaluate(model, full_dataloader, local_rank, args, args.dump_vis if epoch == args.epochs - 1 else False,
                                        coding=True, entropy_model=entropy_model)
            print_str = f'Eval at epoch {epoch + 1} for {hw}: '
            for k, (metric_name, best_v, v) in enumerate(zip(args.metric_names, best_metric_list, results_list)):
                best_v = best_v if best_v > v.max() else v.max()
                if metric_name == 'quant_seen_psnr':
                    psnr_list.append(v.max())
                print_str += f'{metric_name}: {RoundTensor(v, 4)} | '
                best_metric_list[k] = best_v
            print(print_str, flush=True)
            with open('{}/rank0.txt'.format(args.outf), 'a') as f:
                f.write(print_str + '\n')
        torch.save({'epoch': epoch + 1, 'state_dict': model.state_dict(), 'optimizer': optimizer.state_dict()}, '{}/model_latest.pth'.format(args.outf))
        if (epoch + 1) % args.epochs == 0:
            args.cur_epoch = epoch + 1
            args.train_time = str(datetime.now() - start)
            T.Dump2CSV(args, best_metric_list, results_list, psnr_list, f'epoch{epoch + 1}.csv')
    print("Training complete in: " + str(datetime.now() - start) + f"\n Training wo evaluation complete in: {float(sum(time_list))}s")


@torch.no_grad()
def evaluate(model, full_dataloader, local_rank, args, dump_vis=False, coding=False, entropy_model=None):
    """train_nerv_compression.py:457-585: metrics of the de-quantised model (reported in the quant_* slots, as the reference does) and
    the bit accounting.  Symbols are not gathered into a host list (the reference does, to feed its ANS coder): the bit counts
    come from the rate model directly."""
    metric_list = [[] for _ in range(len(args.metric_names))]
    time_list = []
    model.eval()
    model.time_decode = True
    device = next(model.parameters()).device
    n_trans, n_entropy = 0, 0
    for m in _quant_modules(model):
        code_w, quant_w, dequant_w = m.weight_quantizer(m.weight)
        m.dequant_w = dequant_w
        n_trans += sum(p.numel() for p in m.weight_quantizer.parameters())
        if m.bias is not None:
            code_b, quant_b, dequant_b = m.bias_quantizer(m.bias)
            m.dequant_b = dequant_b
            n_trans += sum(p.numel() for p in m.bias_quantizer.parameters())
        if entropy_model is not None:
            m.bitrate_w_dict.update(entropy_model.cal_bitrate(code_w, quant_w, False))
            n_entropy += 2
            if m.bias is not None:
                m.bitrate_b_dict.update(entropy_model.cal_bitrate(code_b, quant_b, False))
                n_entropy += 2
    hnerv = "HNeRV" in args.model
    if hnerv:
        n_trans += sum(p.numel() for p in model.embed_quantizer.parameters())
    e_bits, e_real, e_stats = 0.0, 0.0, 0
    frames_dev = getattr(args, '_frames_dev', None)
    fps, img_data = 0.0, None
    for i, sample in enumerate(full_dataloader):
        norm_idx, img_idx = data_to_gpu(sample['norm_idx'], device), data_to_gpu(sample['idx'], device)
        img_data = frames_dev[img_idx] if frames_dev is not None else data_to_gpu(sample['img'], device)
        img_data, img_gt, inpaint_mask = args.transform_func(img_data, img_idx)
        cur_input = img_data if ('pe' not in args.embed or "HNeRV_Boost" in args.model) else norm_idx
        if hnerv:
            img_embed = model.forward_encoder(cur_input)
            code_e, quant_e, dequant_e = model.forward_embed_quant(img_embed)
            if args.embed_entropy:
                r = entropy_model.cal_bitrate(code_e, quant_e, False)
                e_bits += float(r["bitrate"]); e_real += r["real_bitrate"]; e_stats += 2
            img_out, embed_list, dec_time = model.forward_decoder(dequant_e, norm_idx)
        else:
            img_out, embed_list, dec_time = model(cur_input, norm_idx=norm_idx)
        time_list.append(dec_time)
        pred_psnr, pred_ssim = ops.psnr(img_out, img_gt)[None], ops.msssim(img_out.float(), img_gt)[None]
        for metric_idx, cur_v in enumerate([pred_psnr, pred_ssim]):
            for batch_i, cur_img_idx in enumerate(sample['idx'].tolist()):
                metric_idx_start = 2 if cur_img_idx in args.val_ind_list else 0
                metric_list[metric_idx_start + metric_idx + 4].append(cur_v[:, batch_i])
        if i % args.print_freq == 0 or i == len(full_dataloader) - 1:
            fps = args.batchSize / (sum(time_list) / len(time_list))
            print_str = '[{}] Rank:{}, Eval at Step [{}/{}] , FPS {}, '.format(datetime.now().strftime("%Y/%m/%d %H:%M:%S"), local_rank, i + 1,
                                                                               len(full_dataloader), round(fps, 2))
            for v_name, v_list in zip(args.metric_names, metric_list):
                cur_value = torch.stack(v_list, dim=-1).mean(-1).cpu() if len(v_list) else torch.zeros(1)
                print_str += f'{v_name}: {RoundTensor(cur_value, 4)} | '
            print(print_str, flush=True)
            with open('{}/rank0.txt'.format(args.outf), 'a') as f:
                f.write(print_str + '\n')
    results_list = [torch.stack(v_list, dim=1).mean(1).cpu() if len(v_list) else torch.zeros(1) for v_list in metric_list]
    args.fps = fps
    hw = tuple(img_data.shape[-2:])
    model.time_decode = False
    model.train()
    if coding:
        total_pixels = args.final_size * args.full_data_length
        estimate_bits = float(model.get_bitrate_sum(name="bitrate"))
        data_bits = float(model.get_bitrate_sum(name="real_bitrate"))
        meta_bits = (n_entropy + n_trans) * 32
        if hnerv:
            estimate_bits += e_bits
            data_bits += e_real
            meta_bits += e_stats * 32
        args.total_bpp = (data_bits + meta_bits) / total_pixels
        args.estimate_bpp = (meta_bits + estimate_bits) / total_pixels
        print_str = (f'Gaussian Entropy Model real bpp: {round(args.total_bpp, 6)}, estimated bpp:{round(args.estimate_bpp, 6)}, '
                     f'target_bpp:{round(args.target_bpp, 6)} \n')
        print(print_str, flush=True)
        with open('{}/rank0.txt'.format(args.outf), 'a') as f:
            f.write(print_str + '\n')
    return results_list, hw


if __name__ == '__main__':
    main()
