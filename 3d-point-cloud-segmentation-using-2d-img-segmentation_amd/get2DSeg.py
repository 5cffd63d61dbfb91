"""Drop-in surface of the reference's get2DSeg.py.

The 2D network (OneFormer Swin-L on detectron2) is third-party and not vendored by the reference either; it runs
unchanged on PyTorch-ROCm.  What this module owns is the reference's own post-processing of the logits
(get2DSeg.py:110-118): class = argmax over the 133 classes, pixels whose maximum softmax probability is below
``conf_threshold`` become class 133.  That step is a HIP kernel (f3d_sem_logits_to_mask*), fed straight from the
network's device tensor, so the 133 x H x W logits are read once and only the 1-byte mask is produced.

Two ways out of ``SegmentImage``:
* the reference's: one 8-bit PNG per frame (``<stem>.png``), which ``VotingSegmentation`` reads back;
* ``out_device=...``: the masks stay on the GPU as planes of ONE ``uint8 [V, H, W]`` tensor -- the very tensor
  ``f3d_project_vote_argmax_dev`` samples -- written by ``f3d_sem_logits_to_masks_dev`` on the producer's stream, without
  a synchronisation or a copy to the host per image (no PNG round trip on the fast path).
"""
import glob
import multiprocessing as mp
import os
import random
import sys
from pathlib import Path

import numpy as np

import f3d

# the reference makes its un-vendored ./OneFormer checkout importable (`demo.defaults`, `oneformer`) this way (get2DSeg.py:12)
sys.path.insert(1, os.path.join(sys.path[0], 'OneFormer'))


class OneFormer:
    """Same contract as the reference wrapper (:40-78): ``predict(bgr_uint8[H,W,3]) -> {'sem_seg': [133,H,W] logits, ...}``.
    The predictor is OneFormer's own ``demo.defaults.DefaultPredictor`` (:35) -- the one whose ``__call__`` takes ``task`` --
    not detectron2's ``engine.defaults.DefaultPredictor`` (image only)."""

    def __init__(self, config="./OneFormer/configs/coco/swin/oneformer_swin_large_bs16_100ep.yaml",
                 weights='./OneFormer/PreTrained/COCO/ckpt/150_16_swin_l_oneformer_coco_100ep.pth'):
        print('preparing OneFormer model ...')
        try:
            from detectron2.config import get_cfg
            from detectron2.projects.deeplab import add_deeplab_config
            from oneformer import (add_oneformer_config, add_common_config, add_swin_config, add_dinat_config,
                                   add_convnext_config)
            from demo.defaults import DefaultPredictor
        except ImportError as exc:
            raise ImportError('OneFormer needs detectron2 and the OneFormer checkout (./OneFormer: `oneformer`, `demo.defaults`; not '
                              'vendored by the reference); pass any callable returning [C,H,W] logits to SegmentImage(predictor=...) instead') from exc
        cfg = get_cfg()
        add_deeplab_config(cfg)                                    # the reference's order (:46-52)
        add_common_config(cfg)
        add_swin_config(cfg)
        add_dinat_config(cfg)
        add_convnext_config(cfg)
        add_oneformer_config(cfg)
        cfg.merge_from_file(config)
        cfg.MODEL.WEIGHTS = weights
        self.predictor = DefaultPredictor(cfg)

    def predict(self, image):
        import torch
        with torch.no_grad():
            return self.predictor(image, task='semantic')


def _logits_of(outputs):
    """The 'sem_seg' logits of a predictor's answer (the reference unpacks ``sem, pan, inst = outputs.values()``, :109)."""
    if isinstance(outputs, dict):
        return outputs['sem_seg'] if 'sem_seg' in outputs else next(iter(outputs.values()))
    return outputs


def _launch_stream(torch, device):
    """The producer's stream; the library needs a real stream handle, so the legacy null stream is replaced by a side stream
    ordered behind it (and the null stream made to wait for the side stream afterwards by the caller)."""
    cur = torch.cuda.current_stream(device)
    if cur.cuda_stream != 0:
        return cur, None
    side = torch.cuda.Stream(device)
    side.wait_stream(cur)
    return side, cur


def sem_to_mask(sem, conf_threshold=0.017, low_label=133):
    """[C,H,W] float32 logits (torch CUDA tensor or array) -> uint8 [H,W] class mask on the HOST (reference :110-120)."""
    ctx = f3d.default_context()
    try:
        import torch
        if isinstance(sem, torch.Tensor) and sem.is_cuda:
            out = torch.empty(sem.shape[1:], dtype=torch.uint8, device=sem.device)
            sem_to_mask_device(sem, out, conf_threshold, low_label)
            torch.cuda.current_stream(sem.device).synchronize()
            return out.cpu().numpy()
        if isinstance(sem, torch.Tensor):
            sem = sem.detach().cpu().numpy()
    except ImportError:
        pass
    return ctx.sem_logits_to_mask(np.asarray(sem, np.float32), conf_threshold, low_label)


def sem_to_mask_device(sem, out, conf_threshold=0.017, low_label=133):
    """Device-resident form of the same step: ``sem`` float32 CUDA logits [C,H,W] or [B,C,H,W] -> ``out``, a uint8 CUDA tensor
    [H,W] / [B,H,W] that may be a slice of planes of the [V,H,W] mask tensor the fused call reads.  Enqueued on the current
    stream of ``sem``'s device; returns ``out`` without synchronising and without touching the host."""
    import torch
    if not (isinstance(sem, torch.Tensor) and sem.is_cuda and isinstance(out, torch.Tensor) and out.is_cuda):
        raise ValueError('sem_to_mask_device: sem and out must be CUDA tensors')
    ctx = f3d.default_context(sem.device.index)
    sem = sem.detach()
    if sem.dtype != torch.float32 or not sem.is_contiguous():
        sem = sem.to(torch.float32).contiguous()
    batched = sem.dim() == 4
    b = sem.shape[0] if batched else 1
    c, h, w = sem.shape[-3:]
    if out.dtype != torch.uint8 or not out.is_contiguous() or out.numel() != b * h * w:
        raise ValueError(f'sem_to_mask_device: out must be a contiguous uint8 tensor of {b} x {h} x {w} elements')
    stream, null = _launch_stream(torch, sem.device)
    ctx.sem_logits_to_masks_dev(sem.data_ptr(), b, c, h * w, conf_threshold, low_label, out.data_ptr(), stream.cuda_stream)
    sem.record_stream(stream)
    if null is not None:
        null.wait_stream(stream)
    return out


class DeviceMasks:
    """What ``SegmentImage(..., out_device=...)`` returns: ``masks`` uint8 CUDA tensor [V,H,W] (plane j = frame ``stems[j]``),
    ``kept`` bool CUDA tensor [V] (False = the reference would have skipped the frame: none of ``filter_classes`` occurs in it,
    get2DSeg.py:123-124) and ``written`` (the PNG paths, empty with ``write_png=False``)."""

    def __init__(self, masks, stems, kept, written):
        self.masks, self.stems, self.kept, self.written = masks, stems, kept, written


def _imread(path):
    try:
        import cv2
        return cv2.imread(path)
    except ImportError:
        from PIL import Image
        with Image.open(path) as im:
            return np.asarray(im.convert('RGB'))[:, :, ::-1].copy()


def _imwrite(path, img):
    try:
        import cv2
        cv2.imwrite(path, img)
    except ImportError:
        from PIL import Image
        Image.fromarray(img).save(path)


def _seed_everything(seed=0):
    """get2DSeg.py:83-89."""
    random.seed(seed)
    np.random.seed(seed)
    try:
        import torch as T
        T.manual_seed(seed)
        T.cuda.manual_seed_all(seed)
        T.backends.cudnn.deterministic = True
        T.backends.cudnn.benchmark = False
    except ImportError:
        pass


def _viz(viz_dir, name, image, sem_image):
    """The reference also writes a colour overlay per frame with detectron2's Visualizer (:100-101,121-125); drawn only when
    detectron2 is there (it is absent from this image) -- the mask files never depend on it."""
    try:
        from detectron2.utils.visualizer import Visualizer, ColorMode
        from detectron2.data import MetadataCatalog
    except ImportError:
        return
    v = Visualizer(image[:, :, ::-1], MetadataCatalog.get("coco_2017_val_panoptic"), scale=1.0, instance_mode=ColorMode.IMAGE_BW)
    _imwrite(os.path.join(viz_dir, Path(name).stem + '.png'), v.draw_sem_seg(sem_image).get_image())


def masks_to_device(frames, predictor, conf_threshold=0.017, out=None, low_label=133):
    """The reference's predict -> argmax / softmax-threshold loop (get2DSeg.py:106-120) with the masks kept on the GPU:
    ``frames`` = an iterable of images (whatever ``predictor`` takes: BGR arrays like the reference, or device tensors),
    ``predictor(image)`` -> {'sem_seg': CUDA logits [C,H,W]} (or the logits, or a batch [B,C,H,W] for a batch of frames).
    Returns the uint8 CUDA tensor [V,H,W] (``out`` if given: plane j is written by frame j).  No synchronisation, no D2H."""
    import torch
    j = 0
    for image in frames:
        sem = _logits_of(predictor(image))
        if not (isinstance(sem, torch.Tensor) and sem.is_cuda):
            raise ValueError('masks_to_device: the predictor must return CUDA logits (the reference\'s OneFormer does)')
        b = sem.shape[0] if sem.dim() == 4 else 1
        h, w = sem.shape[-2:]
        if out is None:
            frames_n = len(frames) if hasattr(frames, '__len__') else None
            if frames_n is None:
                raise ValueError('masks_to_device: pass out=[V,H,W] uint8 CUDA tensor for an iterator of unknown length')
            out = torch.empty((frames_n * b, h, w), dtype=torch.uint8, device=sem.device)
        if j + b > out.shape[0] or tuple(out.shape[1:]) != (h, w):
            raise ValueError(f'masks_to_device: frame {j} of {h} x {w} does not fit the mask tensor {tuple(out.shape)}')
        sem_to_mask_device(sem, out[j:j + b], conf_threshold, low_label)
        j += b
    return out


def SegmentImage(input_dir, output_dir, extension="jpg", conf_threshold=0.017, filter_classes=None, predictor=None,
                 out_device=None, write_png=True):
    """One 8-bit class-id PNG per RGB frame, named <stem>.png (reference :82-126).  ``predictor`` (image -> dict with
    'sem_seg' or the logits themselves) defaults to the reference's OneFormer wrapper.

    ``out_device`` (True, or a uint8 CUDA tensor [V,H,W] to fill): the device-resident hand-off -- every frame's mask is written
    straight into its plane of the tensor by the HIP kernel on the producer's stream, nothing is synchronised or copied per frame;
    returns a ``DeviceMasks``.  The PNGs (``write_png``) are then written from ONE copy of the whole stack after the loop."""
    _seed_everything(0)
    mp.set_start_method("spawn", force=True)                       # :91 (the reference never spawns afterwards either)
    filter_classes = set(filter_classes) if filter_classes is not None else None
    os.makedirs(output_dir, exist_ok=True)
    viz_dir = os.path.join(output_dir, 'viz')                      # :100-101
    os.makedirs(viz_dir, exist_ok=True)
    images = sorted(glob.glob(f'{input_dir}/*{extension}'))
    segmentor = predictor if predictor is not None else OneFormer().predict
    print('predicting ...')
    written = []
    if out_device is None or out_device is False:
        for image_path in images:
            image = _imread(image_path)
            mask = sem_to_mask(_logits_of(segmentor(image)), conf_threshold)
            if filter_classes is not None and not (set(np.unique(mask).tolist()) & filter_classes):
                continue
            _viz(viz_dir, os.path.basename(image_path), image, mask)
            out = os.path.join(output_dir, Path(image_path).stem + '.png')
            _imwrite(out, mask)
            written.append(out)
        return written

    import torch
    stack = out_device if isinstance(out_device, torch.Tensor) else None
    kept = []
    flt = None
    for j, image_path in enumerate(images):
        sem = _logits_of(segmentor(_imread(image_path)))
        if stack is None:
            stack = torch.empty((len(images),) + tuple(sem.shape[-2:]), dtype=torch.uint8, device=sem.device)
        sem_to_mask_device(sem, stack[j], conf_threshold)
        if filter_classes is not None:                             # :123-124, decided on the device; read once after the loop
            if flt is None:
                flt = torch.tensor(sorted(filter_classes), dtype=torch.uint8, device=stack.device)
            kept.append(torch.isin(stack[j], flt).any())
    stems = [Path(p).stem for p in images]
    if stack is None:
        return DeviceMasks(None, stems, None, written)
    kept_t = torch.stack(kept) if kept else torch.ones(len(images), dtype=torch.bool, device=stack.device)
    if write_png and images:
        host, keep = stack[:len(images)].cpu().numpy(), kept_t.cpu().numpy()
        for j, image_path in enumerate(images):
            if keep[j]:
                out = os.path.join(output_dir, stems[j] + '.png')
                _imwrite(out, host[j])
                written.append(out)
    return DeviceMasks(stack, stems, kept_t, written)
