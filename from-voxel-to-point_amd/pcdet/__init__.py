"""MI355X-native replacement of the `pcdet.ops` hot path of jialeli1/From-Voxel-to-Point.

Only the hot-path boundary lives here (pcdet.ops.*, pcdet.datasets.processor.voxel_generator,
pcdet.utils.spconv_utils); the detectors, datasets and tools of the reference are consumers of
this boundary and are not re-implemented (see INTEGRATION.md for how to overlay this tree).
"""
__version__ = "0.3.0+fv2p.mi355x"
