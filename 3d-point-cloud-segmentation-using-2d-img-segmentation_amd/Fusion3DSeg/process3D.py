"""Call surface of the reference's Fusion3DSeg/process3D.py: fuse the frames of a capture into one sparse cloud and write
the fusion directory that get3DSeg.segment continues from."""
import os
import time
from pathlib import Path

from Fusion3DSeg.fusion import Fusion


def process3DSeg(input_data_path, output_path, radius=0.05, angle=10, stride=10, point_range=(0.1, 4), decimation=1,
                 min_occ=3, verbose=False):
    """reference process3D.py:14-68 -> (points, normals, colours, nmerges, occurences, nframes, depth_hw, adj).

    Like the reference, the ``min_occ`` filter is evaluated on ``nmerges`` and its result is then discarded (quirk Q8): the
    unfiltered cloud is what gets written and returned."""
    merged = os.path.join(input_data_path, 'PointcloudMergeResults')
    if os.path.exists(merged):
        stem = [f for f in os.listdir(merged) if 'tofsegment' in f][0][:-4]
        suffix = stem.split('_', 1)[1]
    else:
        print('tofcameradata not found')                                      # the reference then fails with a NameError as well
    tof = os.path.join(merged, f'tofsegment_{suffix}.pkl')
    rts = os.path.join(merged, f'rtscameradata_{suffix}.pkl')
    t0 = time.perf_counter()
    fuser = Fusion(tof, rts, point_range, decimation)
    pts, nrm, clr, nmerges, occurences = fuser.fuse(radius, angle, stride, point_range[1], skip=1, verbose=verbose)
    if verbose:
        print(f'\ntotal {fuser.npts * fuser.nframes} points from {fuser.nframes} frames are fused into {len(pts)} points')
        print(f'time taken for fusion = {(time.perf_counter() - t0) / 60} minutes')
    if min_occ is not None:
        mask, _ = fuser.filter(nmerges, min_occ, [pts, nrm, clr, nmerges, occurences], less_than=False)
        if verbose:
            print(f'remaining points after frame occurence thresholding with {min_occ} = {mask.sum()}')
    fuser.dump_data(Path(output_path), pts, nrm, clr, nmerges, occurences, True, verbose)
    return tuple(fuser.load_data(Path(output_path)))
