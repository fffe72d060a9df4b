"""TEST INFRASTRUCTURE -- argument namespaces of the BASELINE configs and tiny variants.

The values are the flags of the reference's recipes (scripts/regression/bunny/nerv_boost.sh:4-9,
scripts/regression/UVG/hnerv_boost.sh:7-12, scripts/regression/UVG/enerv_boost.sh:7-12) with the derived
``fc_dim`` the reference's size solver produces (train_nerv_all.py:193-217): C1 -> 30, C3 -> 95, C4 -> 59.
"""
from types import SimpleNamespace


def _base(**kw):
    d = dict(embed="pe_1.25_80", lfreq="pi", fc_hw="9_16", fc_dim=None, ch_t=32, ks="0_3_3", enc_blks=1, enc_strds=[],
             enc_dim="64_16", dec_strds=[5, 2, 2, 2, 2], dec_blks=[1, 1, 2, 2, 2], reduce=2, lower_width=12,
             conv_type=["convnext", "pshuffel_3x3"], norm="none", act="sin", sft_block="res_sft", out_bias="tanh",
             outf="unit", quant=False, block_dim=128, model="NeRV_Boost")
    d.update(kw)
    return SimpleNamespace(**d)


def c1():      # NeRV-boost 1.5M, Bunny 720x1280
    return _base(fc_dim=30)


def c3():      # HNeRV-boost 3M, 1080x1920
    return _base(model="HNeRV_Boost", enc_strds=[5, 3, 2, 2, 2], enc_dim="64_16", dec_strds=[5, 3, 2, 2, 2], ks="0_1_5",
                 reduce=1.2, fc_dim=95)


def c4():      # E-NeRV-boost 3M, 1080x1920
    return _base(model="ENeRV_Boost", dec_strds=[5, 3, 2, 2, 2], fc_dim=59, block_dim=128)


def tiny_nerv():   # 9x16 -> 45x80 -> 90x160 -> 180x320 (+1 stride-1 block); > 160 on the short side for MS-SSIM
    return _base(fc_dim=8, dec_strds=[5, 2, 2], dec_blks=[1, 1, 2], lower_width=6)


def tiny_hnerv():  # 180x320 input, encoder strides 5,2,2 (x20) -> 9x16 embedding; decoder 5,2,2
    return _base(model="HNeRV_Boost", enc_strds=[5, 2, 2], enc_dim="16_4", dec_strds=[5, 2, 2], dec_blks=[1, 1, 2],
                 ks="0_1_5", reduce=1.2, lower_width=6, fc_dim=10)


def tiny_hnerv_quant():  # the same model built for the CEM compression path (scripts/compression/hnerv_boost.sh quantiser flags)
    a = tiny_hnerv()
    a.__dict__.update(quant=True, quant_model_bit=8, quant_bias_bit=8, quant_embed_bit=8, per_channel_w=False, per_channel_b=False,
                      per_channel_e=False, quantizer_w="scale", quantizer_b="scale", quantizer_e="scalebeta", embed_entropy=True)
    return a


def c5():      # C3's model built for the CEM compression path (scripts/compression/hnerv_boost.sh:7-16 quantiser flags), BASELINE configs[4]
    a = c3()
    a.__dict__.update(quant=True, quant_model_bit=8, quant_bias_bit=8, quant_embed_bit=8, per_channel_w=False, per_channel_b=False,
                      per_channel_e=False, quantizer_w="scale", quantizer_b="scale", quantizer_e="scalebeta", embed_entropy=True)
    return a


def tiny_enerv():
    return _base(model="ENeRV_Boost", dec_strds=[5, 2, 2], dec_blks=[1, 1, 2], fc_dim=8, lower_width=6, block_dim=64)
