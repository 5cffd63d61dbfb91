"""Drop-in surface of the reference's get2DSeg.py.

The 2D network (OneFormer Swin-L on detectron2) is third-party and not vendored by the reference either; it runs
unchanged on PyTorch-ROCm.  What this module owns is the reference's own post-processing of the logits
(get2DSeg.py:110-118): class = argmax over the 133 classes, pixels whose maximum softmax probability is below
``conf_threshold`` become class 133.  That step is a HIP kernel (f3d_sem_logits_to_mask*), fed straight from the
network's device tensor, so the 133 x H x W logits are read once and only the 1-byte mask leaves the GPU.
"""
import glob
import os
from pathlib import Path

import numpy as np

import f3d


class OneFormer:
    """Same contract as the reference wrapper (:40-78): ``predict(bgr_uint8[H,W,3]) -> {'sem_seg': [133,H,W] logits, ...}``."""

    def __init__(self, config="./OneFormer/configs/coco/swin/oneformer_swin_large_bs16_100ep.yaml",
                 weights='./OneFormer/PreTrained/COCO/ckpt/150_16_swin_l_oneformer_coco_100ep.pth'):
        try:
            from detectron2.config import get_cfg
            from detectron2.projects.deeplab import add_deeplab_config
            from detectron2.engine.defaults import DefaultPredictor
            from oneformer import (add_oneformer_config, add_common_config, add_swin_config, add_dinat_config,
                                   add_convnext_config)
        except ImportError as exc:
            raise ImportError('OneFormer needs detectron2 and the OneFormer package (not vendored by the reference); '
                              'pass any callable returning [C,H,W] logits to SegmentImage(predictor=...) instead') from exc
        cfg = get_cfg()
        for add in (add_deeplab_config, add_common_config, add_swin_config, add_dinat_config, add_convnext_config, add_oneformer_config):
            add(cfg)
        cfg.merge_from_file(config)
        cfg.MODEL.WEIGHTS = weights
        self.predictor = DefaultPredictor(cfg)

    def predict(self, image):
        import torch
        with torch.no_grad():
            return self.predictor(image, task='semantic')


def sem_to_mask(sem, conf_threshold=0.017, low_label=133):
    """[C,H,W] float32 logits (torch CUDA tensor or array) -> uint8 [H,W] class mask (reference :110-118)."""
    ctx = f3d.default_context()
    try:
        import torch
        if isinstance(sem, torch.Tensor) and sem.is_cuda:
            sem = sem.detach().to(torch.float32).contiguous()
            c, h, w = sem.shape
            out = torch.empty((h, w), dtype=torch.uint8, device=sem.device)
            stream = torch.cuda.current_stream(sem.device)
            ev = None
            if stream.cuda_stream == 0:                          # the library needs a real stream handle; order it behind the producer
                side = torch.cuda.Stream(sem.device)
                side.wait_stream(stream)
                stream = side
            ctx.sem_logits_to_mask_dev(sem.data_ptr(), c, h * w, conf_threshold, low_label, out.data_ptr(), stream.cuda_stream)
            stream.synchronize()
            return out.cpu().numpy()
        if isinstance(sem, torch.Tensor):
            sem = sem.detach().cpu().numpy()
    except ImportError:
        pass
    return ctx.sem_logits_to_mask(np.asarray(sem, np.float32), conf_threshold, low_label)


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


def SegmentImage(input_dir, output_dir, extension="jpg", conf_threshold=0.017, filter_classes=None, predictor=None):
    """One 8-bit class-id PNG per RGB frame, named <stem>.png (reference :82-126).  ``predictor`` (image -> dict with
    'sem_seg' or the logits themselves) defaults to the reference's OneFormer wrapper."""
    filter_classes = set(filter_classes) if filter_classes is not None else None
    os.makedirs(output_dir, exist_ok=True)
    segmentor = predictor if predictor is not None else OneFormer().predict
    written = []
    for image_path in sorted(glob.glob(f'{input_dir}/*{extension}')):
        outputs = segmentor(_imread(image_path))
        sem = outputs['sem_seg'] if isinstance(outputs, dict) else outputs
        mask = sem_to_mask(sem, conf_threshold)
        if filter_classes is not None and not (set(np.unique(mask).tolist()) & filter_classes):
            continue
        out = os.path.join(output_dir, Path(image_path).stem + '.png')
        _imwrite(out, mask)
        written.append(out)
    return written
