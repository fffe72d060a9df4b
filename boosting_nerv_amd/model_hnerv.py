"""HNeRV_Boost -- host-side mirror of the reference's model_hnerv.py:178-322.  The ConvNeXt content encoder runs on
stock PyTorch-ROCm ops; the conditional decoder (stem_t, decoder[0..], head) runs on the HIP kernels."""
import time

import torch
import torch.nn as nn

from .model_blocks import *  # noqa: F401,F403
from .model_blocks import ConvNeXt, CustomConv2d, NeRV_MLP, NeRVBlock, PositionEncoding, head_out
from .lib.transform_ops import quant_map
from .model_nerv import _CEMHooks, decoder_layers_forward


class HNeRV(nn.Module):
    def __init__(self, args):
        raise NotImplementedError("HNeRV (non-boost baseline, model_hnerv.py:11-158) is outside the conditional-decoder path (SURVEY section 2)")


class HNeRVDecoder(nn.Module):
    def __init__(self, model):
        raise NotImplementedError("HNeRVDecoder (non-boost baseline helper, model_hnerv.py:160-175) is outside the path (SURVEY section 2)")


class HNeRV_Boost(_CEMHooks, nn.Module):
    lazy_flush_ok = True     # (engine.TrainStep: deferred slab reductions are flushed by their first reader; all readers are this package's operators)
    def __init__(self, args):
        super().__init__()
        self.embed = args.embed
        ks_enc, ks_dec1, ks_dec2 = [int(x) for x in args.ks.split("_")]
        enc_blks = args.enc_blks
        enc_dim1, enc_dim2 = [int(x) for x in args.enc_dim.split("_")]
        c_out_list = [enc_dim1] * len(args.enc_strds)
        c_out_list[-1] = enc_dim2
        self.encoder = ConvNeXt(stage_blocks=enc_blks, strds=args.enc_strds, dims=c_out_list, drop_path_rate=0)

        self.pe_embed_t = PositionEncoding(args.embed, args.lfreq)
        mlp_dim_list = [int(self.pe_embed_t.embed_length)] + [int(args.ch_t * 2)] + [args.ch_t]
        self.stem_t = NeRV_MLP(dim_list=mlp_dim_list, bias=True, act=args.act, omega=1, args=args)

        decoder_layers = []
        ngf = args.fc_dim
        decoder_layers.append(NeRVBlock(dec_block=False, conv_type="conv", ngf=enc_dim2, new_ngf=ngf, ks=0, strd=1, bias=True,
                                        norm=args.norm, act=args.act, sft_ngf=args.ch_t, args=args))
        for i, strd in enumerate(args.dec_strds):
            reduction = sqrt(strd) if args.reduce == -1 else args.reduce
            new_ngf = int(max(round(ngf / reduction), args.lower_width))
            for j in range(args.dec_blks[i]):
                decoder_layers.append(NeRVBlock(dec_block=True, conv_type=args.conv_type[1], ngf=ngf, new_ngf=new_ngf,
                                                ks=min(ks_dec1 + 2 * i, ks_dec2), strd=1 if j else strd, bias=True, norm=args.norm,
                                                act=args.act, sft_ngf=args.ch_t, args=args))
                ngf = new_ngf
        self.decoder = nn.ModuleList(decoder_layers)
        self.head_layer = CustomConv2d(ngf, 3, 3, 1, 1, args=args)
        self.out_bias = args.out_bias
        if args.quant:                                         # model_hnerv.py:216-220
            self.embed_quantizer = quant_map[args.quantizer_e](args.quant_embed_bit, signed=False, per_channel=args.per_channel_e)
            self.bitrate_e_dict = {}
        else:
            self.embed_quantizer = None
        self.outf = args.outf
        self.time_decode = False

    def _decode(self, img_embed, norm_idx):
        embed_list = [img_embed]
        dec_start = time.time()
        # norm_idx arrives as float64 from the loader: the PE product and sin/cos are evaluated in fp64 and cast (:241)
        t_embed = self.stem_t(self.pe_embed_t(norm_idx[:, None]).float())
        output = decoder_layers_forward(self.decoder, img_embed, t_embed, embed_list)
        img_out = head_out(self.head_layer, output, self.out_bias)
        if self.time_decode and torch.cuda.is_available():
            torch.cuda.synchronize()
        return img_out, embed_list, time.time() - dec_start

    def forward(self, input, input_embed=None, entropy_model=None, pre_img=None, post_img=None, norm_idx=None):
        img_embed = input_embed if input_embed is not None else self.encoder(input)
        if self.embed_quantizer is not None:                   # model_hnerv.py:230-234
            self.embed_quantizer.init_data(img_embed)
            code_e, quant_e, img_embed = self.embed_quantizer(img_embed)
            if entropy_model is not None:
                self.bitrate_e_dict.update(entropy_model.cal_bitrate(code_e, quant_e, self.training))
        if pre_img is not None and post_img is not None:
            img_embed = 0.5 * (self.encoder(pre_img) + self.encoder(post_img))
        return self._decode(img_embed, norm_idx)

    def forward_encoder(self, input):
        return self.encoder(input)

    def forward_embed_quant(self, img_embed, entropy_model=None):      # model_hnerv.py:256-260
        code, quant, img_embed = self.embed_quantizer(img_embed)
        if entropy_model is not None:
            self.bitrate_e_dict.update(entropy_model.cal_bitrate(code, quant, self.training))
        return code, quant, img_embed

    def forward_decoder(self, img_embed, norm_idx):
        return self._decode(img_embed, norm_idx)

    def decoder_params(self):
        return (sum([p.data.nelement() for p in self.parameters()]) - sum([p.data.nelement() for p in self.encoder.parameters()])) / 1e6
