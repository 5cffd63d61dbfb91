"""Drop-in for the reference's Fusion3DSeg/segUtils/voting.py (class VotingSegmentation).

Same constructor, attributes, methods, return types and quirks (reference file:line in each
method); the scatter vote and the segmentation run as HIP kernels (f3d_vote_uv2pt*, f3d_segment_votes*).
The vote matrix stays on the GPU for the whole ``vote()`` loop and is downloaded once.
OpenCV is optional: masks are read with cv2 when it is importable, else with Pillow, and the
nearest-neighbour resize is done here (same index rule as cv2.INTER_NEAREST).
"""
from pathlib import Path

import numpy as np

import f3d


def _imread_gray(path):
    """cv2.imread(path, 0) equivalent for 8-bit label PNGs (reference voting.py:66)."""
    try:
        import cv2
        return cv2.imread(str(path), 0)
    except ImportError:
        from PIL import Image
        with Image.open(path) as im:
            return np.asarray(im.convert('L'), dtype=np.uint8)


def resize_nearest(mask, w, h):
    """cv2.resize(mask, (w, h), interpolation=INTER_NEAREST): src index = min(floor(dst * src/dst), src-1)."""
    sh, sw = mask.shape[:2]
    if (sh, sw) == (h, w):
        return mask
    xs = np.minimum(np.floor(np.arange(w) * (1.0 / (w / sw))).astype(np.int64), sw - 1)
    ys = np.minimum(np.floor(np.arange(h) * (1.0 / (h / sh))).astype(np.int64), sh - 1)
    return mask[ys[:, None], xs[None, :]]


class VotingSegmentation:
    """Voting based 3D point-cloud segmentation from 2D masks and uv2pt lookups (reference voting.py:11-137)."""

    def __init__(self, npts, depth_hw, maskdir, uv2ptdir, nclasses, votes_file=None):
        if votes_file is None:
            self.npts = npts
            self.depth_hw = depth_hw
            self.nclasses = nclasses
            self.votes = np.zeros((npts, nclasses + 1))
            self.mask_files, self.uv2pt_files = self._get_filenames(maskdir, uv2ptdir)
            self.nframes = len(self.mask_files)
        else:                                            # quirk Q2: nclasses becomes the column count (reference :39-40)
            self.votes = np.load(votes_file)
            self.nclasses = self.votes.shape[1]

    def _get_filenames(self, maskdir, uv2ptdir):
        """Frames present in both directories, paired by stem (reference :42-54)."""
        maskdir, uv2ptdir = Path(maskdir), Path(uv2ptdir)
        masks = {p.stem: p for p in maskdir.iterdir() if p.is_file()}
        luts = {p.stem: p for p in uv2ptdir.iterdir() if p.is_file()}
        mask_ext = next(iter(masks.values())).suffix
        lut_ext = next(iter(luts.values())).suffix
        common = set(masks) & set(luts)
        return ([(maskdir / s).with_suffix(mask_ext) for s in common],
                [(uv2ptdir / s).with_suffix(lut_ext) for s in common])

    def _read_data(self, idx):
        return _imread_gray(self.mask_files[idx]), np.load(self.uv2pt_files[idx])

    def zero(self):
        self.votes = np.zeros_like(self.votes)

    def vote(self, resize=True, verbose=False, filename=None):
        """Accumulate the votes of every frame (reference :75-104); returns float64 [npts, nclasses+1].

        Per frame ``votes[uv2pt[valid], mask[valid]] += 1`` with NumPy's buffered semantics: a (point, label)
        pair adds 1 per frame however many pixels map to it (quirk Q1); an out-of-range label or point raises
        IndexError before the frame changes anything.
        """
        h, w = self.depth_hw
        ctx = f3d.default_context()
        session = _DeviceVotes(ctx, self.votes)
        if verbose:
            print('voting ... ')
        # frames are read on the host and handed to the GPU a batch at a time: one upload of all lookups and masks, one
        # kernel pair for the batch (f3d_vote_uv2pt_batch*) instead of two copies and three launches per frame
        batch_l, batch_m, pending_error = [], [], None
        for i in range(self.nframes):
            if verbose:
                print(f'frame/total = {i + 1}/{self.nframes}, progress = {((i + 1) * 100 / self.nframes):.3}%')
            mask, uv2pt = self._read_data(i)
            mask = resize_nearest(mask, w, h) if resize else mask
            mask = np.ascontiguousarray(mask, dtype=np.uint8).reshape(-1)
            lut = np.asarray(uv2pt, dtype=np.int32).reshape(-1)
            if len(lut) != len(mask):                       # NumPy raises at this frame; the earlier ones stay applied
                pending_error = IndexError(f'boolean index did not match: uv2pt has {len(lut)} entries, the mask {len(mask)}')
                break
            if len(lut) != h * w:                           # a lookup of another size than depth_hw: its own single-frame call
                session.add_frames(batch_l, batch_m, h, w); batch_l, batch_m = [], []
                session.add_frame(lut, mask)
                continue
            batch_l.append(lut); batch_m.append(mask)
            if len(batch_l) * h * w >= 1 << 26:             # ~64 frames of 1024^2: bound the host staging memory
                session.add_frames(batch_l, batch_m, h, w); batch_l, batch_m = [], []
        session.add_frames(batch_l, batch_m, h, w)
        try:
            self.votes = session.download()
            if pending_error is not None:
                raise pending_error
        except IndexError:
            self.votes = session.download(check=False)      # frames before the offending one stay applied, as in the reference
            raise
        if filename is not None:
            Path(filename).parent.mkdir(exist_ok=True, parents=True)
            np.save(filename, self.votes)
        return self.votes

    def segment(self, threshold=0.5, filter_classes=None, votes=None):
        """Per-point class from the votes (reference :106-137) -> int64 [npts].

        argmax over all columns or over ``votes[:, filter_classes]`` (first maximum wins), ``nclasses`` for points
        with no votes, with max/total < threshold or with a zero maximum; then the reference's sequential
        index->class remap (which aliases when a class id is smaller than the list length, quirk Q3).
        """
        votes = self.votes if votes is None else votes
        votes = self.vote() if votes is None else votes
        return f3d.default_context().segment_votes(votes, self.nclasses, threshold, filter_classes)


class _DeviceVotes:
    """Keeps the vote matrix resident on the GPU across frames when torch is available (device memory plumbing);
    otherwise every frame goes through the host-pointer entry point (still the HIP kernels)."""

    def __init__(self, ctx, votes):
        self.ctx, self.host = ctx, np.ascontiguousarray(votes, dtype=np.float64)
        self.torch, self.error = None, None
        try:
            import torch
            if torch.cuda.is_available():
                self.torch = torch
                self.dev = torch.device('cuda', ctx.device)
                self.t = torch.from_numpy(self.host).to(self.dev)
                self.stream = torch.cuda.Stream(self.dev)
        except ImportError:
            pass

    def add_frames(self, luts, masks, h, w):
        """Frames of h*w lookups each, in order, as ONE batched call."""
        if not luts:
            return
        lut = np.ascontiguousarray(np.stack(luts), dtype=np.int32)
        m = np.ascontiguousarray(np.stack(masks), dtype=np.uint8)
        if self.torch is None:
            if self.error is not None:                                  # the reference stopped at the offending frame
                return
            try:
                self.ctx.vote_uv2pt_batch(self.host, lut, m, h, w)
            except IndexError as exc:                                   # frames before the offending one are applied, like NumPy
                self.error = self.error or exc
            return
        torch = self.torch
        with torch.cuda.stream(self.stream):
            dl = torch.from_numpy(lut).to(self.dev)
            dm = torch.from_numpy(m).to(self.dev)
            self.ctx.vote_uv2pt_batch_dev(dl.data_ptr(), dm.data_ptr(), len(lut), h, w, self.t.data_ptr(), self.t.shape[0], self.t.shape[1],
                                          self.stream.cuda_stream)

    def add_frame(self, uv2pt, mask_flat):
        lut = np.array(uv2pt, dtype=np.int32).reshape(-1)               # private, writable copies
        m = np.array(mask_flat, dtype=np.uint8).reshape(-1)
        if len(lut) != len(m):
            raise IndexError(f'boolean index did not match: uv2pt has {len(lut)} entries, the mask {len(m)}')
        if self.torch is None:
            if self.error is None:
                try:
                    self.ctx.vote_uv2pt(self.host, lut, m)
                except IndexError as exc:
                    self.error = exc
            return
        torch = self.torch
        with torch.cuda.stream(self.stream):
            dl = torch.from_numpy(lut).to(self.dev)
            dm = torch.from_numpy(m).to(self.dev)
            self.ctx.vote_uv2pt_dev(dl.data_ptr(), dm.data_ptr(), len(lut), self.t.data_ptr(), self.t.shape[0], self.t.shape[1],
                                    self.stream.cuda_stream)
            # no per-frame synchronisation: an out-of-range index sets a sticky device flag, the offending frame and every
            # later one write nothing, and download() raises the IndexError -- the same votes state as the reference's
            # exception at that frame (voting.py:98)

    def download(self, check=True):
        if self.torch is None:
            if check and self.error is not None:
                raise self.error
            return self.host
        self.stream.synchronize()
        if check:
            self.ctx.take_device_error(self.stream.cuda_stream)     # IndexError for the first bad frame, if any
        return self.t.cpu().numpy()
