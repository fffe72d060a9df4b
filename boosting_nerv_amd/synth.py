"""Deterministic synthetic video source: no dataset ships with the reference and the GPU box has no
network, so BASELINE's "Bunny-shaped" (132 x 3x720x1280) and "UVG-shaped" (600 x 3x1080x1920) clips are generated.

frame n, channel c:  sum_{k=1..8} A_k sin(2 pi (fx_k x/W + fy_k y/H + ft_k n/N) + phi_{k,c}),  A_k ~ 1/k,
plus a static seeded noise texture (H/8 x W/8, bilinearly upsampled, weight 0.15), mapped affinely to [0,1], rounded
to uint8 and divided by 255 (mimics PNG -> ToTensor of the reference's VideoDataSet, hnerv_utils.py:23,46)."""
import math

import torch

SEED = 20240418


class SyntheticVideo:
    def __init__(self, n_frames, height, width, seed=SEED):
        self.n, self.h, self.w = int(n_frames), int(height), int(width)
        g = torch.Generator().manual_seed(seed)
        K = 8
        self.fx = torch.randint(1, 25, (K,), generator=g).float()
        self.fy = torch.randint(1, 25, (K,), generator=g).float()
        self.ft = torch.randint(0, 4, (K,), generator=g).float()
        self.phi = torch.rand(K, 3, generator=g) * 2 * math.pi
        self.amp = 1.0 / torch.arange(1, K + 1).float()
        noise = torch.rand(1, 3, max(self.h // 8, 2), max(self.w // 8, 2), generator=g)
        self.texture = torch.nn.functional.interpolate(noise, size=(self.h, self.w), mode="bilinear", align_corners=False)[0]
        self.norm = float(self.amp.sum())

    def __len__(self):
        return self.n

    def frame(self, idx, device="cpu"):
        """[3,H,W] float32 in [0,1] on an 8-bit grid."""
        y = torch.arange(self.h, dtype=torch.float32, device=device)[:, None] / self.h
        x = torch.arange(self.w, dtype=torch.float32, device=device)[None, :] / self.w
        out = torch.zeros(3, self.h, self.w, dtype=torch.float32, device=device)
        for k in range(self.amp.numel()):
            base = 2 * math.pi * (self.fx[k].item() * x + self.fy[k].item() * y + self.ft[k].item() * idx / self.n)
            for c in range(3):
                out[c] += self.amp[k].item() * torch.sin(base + self.phi[k, c].item())
        out = out / (2 * self.norm) + 0.5
        out = 0.85 * out + 0.15 * self.texture.to(device)
        return torch.round(out.clamp(0, 1) * 255.0) / 255.0


def dump_png(directory, n_frames, height, width, device="cpu"):
    """Write the clip as frame_%04d.png files: the layout the reference's VideoDataSet reads (hnerv_utils.py:19-47; sorted directory
    listing, PNG -> ToTensor), so the loader path can be driven with real files.  The frames are on an 8-bit grid, so the PNGs hold
    them exactly."""
    import os
    from PIL import Image
    os.makedirs(directory, exist_ok=True)
    vid = SyntheticVideo(n_frames, height, width)
    for i in range(n_frames):
        arr = (vid.frame(i, device=device) * 255.0).round().clamp(0, 255).to(torch.uint8).permute(1, 2, 0).cpu().numpy()
        Image.fromarray(arr).save(os.path.join(directory, f"frame_{i:04d}.png"))
    return directory


def parse_spec(spec):
    """'synthetic:bunny' | 'synthetic:uvg' | 'synthetic:NxHxW'"""
    s = spec.split(":", 1)[1].lower()
    if s == "bunny":
        return 132, 720, 1280
    if s == "uvg":
        return 600, 1080, 1920
    n, h, w = [int(v) for v in s.split("x")]
    return n, h, w
