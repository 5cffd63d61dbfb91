"""Call surface of the reference's Fusion3DSeg/process3D.py.

``process3DSeg`` drives ``Fusion.fuse`` (greedy patch merging, reference fusion.py:134-324), which is row (f)#2 of
the scope table and not part of this round; the function exists so that callers fail with a clear message instead
of an ImportError.  Everything downstream of an existing fusion directory (voting, segmentation, box merge) is
available through get3DSeg.
"""


def process3DSeg(input_data_path, output_path, radius=0.05, angle=10, stride=10, point_range=(0.1, 4), decimation=1,
                 min_occ=3, verbose=False):
    raise NotImplementedError('process3DSeg needs Fusion.fuse (reference fusion.py:134-324), which this round does not '
                              'provide; run the reference fusion once and continue with get3DSeg.segment on its output')
