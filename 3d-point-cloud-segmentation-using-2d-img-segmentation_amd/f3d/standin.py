"""A STAND-IN for the 2D segmentation network, for bench.py's `c3_end_to_end` leg and the hand-off tests only.

The reference's backbone is OneFormer Swin-L on detectron2 (get2DSeg.py:40-78); neither the code nor the weights are vendored by
the reference, and neither is in this image (no network).  What the end-to-end leg needs from it is its CONTRACT -- a PyTorch-ROCm
module that turns a BGR uint8 frame into float32 logits ``[133, H, W]`` resident on the GPU -- so that the part this build owns
(logits -> mask kernel -> the [V,H,W] device tensor -> fused projection / vote / segment) is exercised exactly as a real
backbone would feed it.  This module is that contract with random weights: a patch embedding (8 x 8 patches), a few MLP blocks
and a 133-way head (all GEMMs: rocBLAS / hipBLASLt on the MFMA units, bf16), bilinear upsampling to the frame size.  It says
nothing about OneFormer's speed or accuracy; a real Swin-L forward is two orders of magnitude more work per frame.
"""
import torch
import torch.nn.functional as F


class StandInSegNet(torch.nn.Module):
    def __init__(self, nclasses=133, patch=8, width=256, depth=2, seed=0, logit_scale=0.5, dtype=torch.bfloat16):
        super().__init__()
        g = torch.Generator().manual_seed(seed)
        self.patch, self.nclasses, self.logit_scale, self.dtype = patch, nclasses, logit_scale, dtype

        def lin(i, o):
            layer = torch.nn.Linear(i, o)
            with torch.no_grad():
                layer.weight.copy_(torch.randn(o, i, generator=g) / i ** 0.5)
                layer.bias.zero_()
            return layer
        self.embed = lin(3 * patch * patch, width)
        self.blocks = torch.nn.ModuleList([torch.nn.ModuleList([torch.nn.LayerNorm(width), lin(width, 4 * width), lin(4 * width, width)])
                                           for _ in range(depth)])
        self.norm = torch.nn.LayerNorm(width)
        self.head = lin(width, nclasses)
        self.to(dtype)

    @torch.no_grad()
    def forward(self, frames):
        """frames: uint8 CUDA tensor [B,H,W,3] (BGR like cv2.imread) or [H,W,3]; H, W multiples of the patch size.
        Returns float32 logits [B,nclasses,H,W] ([nclasses,H,W] for a single frame)."""
        single = frames.dim() == 3
        x = frames[None] if single else frames
        b, h, w, _ = x.shape
        p = self.patch
        x = (x.to(self.dtype) / 255.0 - 0.5).reshape(b, h // p, p, w // p, p, 3).permute(0, 1, 3, 2, 4, 5).reshape(b, (h // p) * (w // p), 3 * p * p)
        x = self.embed(x)
        for norm, up, down in self.blocks:
            x = x + down(F.gelu(up(norm(x))))
        x = self.head(self.norm(x)) * self.logit_scale
        x = x.reshape(b, h // p, w // p, self.nclasses).permute(0, 3, 1, 2).contiguous().float()   # small; NCHW so that the upsampling writes NCHW directly
        x = F.interpolate(x, size=(h, w), mode='bilinear', align_corners=False)
        return x[0] if single else x

    def predict(self, image):
        """The reference wrapper's contract (get2DSeg.py:60-78): BGR uint8 image [H,W,3] (array or CUDA tensor) ->
        {'sem_seg': logits [133,H,W] on the GPU, ...}."""
        dev = next(self.parameters()).device
        if not isinstance(image, torch.Tensor):
            image = torch.from_numpy(image)
        return {'sem_seg': self.forward(image.to(dev)), 'panoptic_seg': None, 'instances': None}


def synthetic_frames(v, h, w, device, seed=4321):
    """uint8 [v,h,w,3] BGR frames, seeded, generated on the device: smooth colour fields plus noise (so that the stand-in's
    argmax changes slowly across a frame, like a real segmentation, instead of per pixel)."""
    g = torch.Generator(device=device).manual_seed(seed)
    coarse = torch.rand((v, 3, max(h // 64, 2), max(w // 64, 2)), generator=g, device=device)
    img = F.interpolate(coarse, size=(h, w), mode='bilinear', align_corners=False)
    img = img + 0.05 * torch.rand((v, 3, h, w), generator=g, device=device)
    return (img.clamp(0, 1) * 255).to(torch.uint8).permute(0, 2, 3, 1).contiguous()
