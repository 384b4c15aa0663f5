"""Box helpers the ops layer imports from the reference's pcdet/utils/box_utils.py
(enlarge_box3d :190-203, boxes_to_corners_3d :28-53)."""
from . import common_utils


def enlarge_box3d(boxes3d, extra_width=(0, 0, 0)):
    boxes3d, is_numpy = common_utils.check_numpy_to_torch(boxes3d)
    large = boxes3d.clone()
    large[:, 3:6] += boxes3d.new_tensor(extra_width)[None, :]
    return large


def boxes_to_corners_3d(boxes3d):
    boxes3d, is_numpy = common_utils.check_numpy_to_torch(boxes3d)
    template = boxes3d.new_tensor(([1, 1, -1], [1, -1, -1], [-1, -1, -1], [-1, 1, -1],
                                   [1, 1, 1], [1, -1, 1], [-1, -1, 1], [-1, 1, 1])) / 2
    corners = boxes3d[:, None, 3:6].repeat(1, 8, 1) * template[None, :, :]
    corners = common_utils.rotate_points_along_z(corners.view(-1, 8, 3), boxes3d[:, 6]).view(-1, 8, 3)
    corners += boxes3d[:, None, 0:3]
    return corners.numpy() if is_numpy else corners


def expand_box3d(boxes3d, expand_times=0.6):
    """Scales the box size by (1 + expand_times) (reference box_utils.py:205-227)."""
    boxes3d, is_numpy = common_utils.check_numpy_to_torch(boxes3d)
    large = boxes3d.clone()
    large[:, 3:6] += boxes3d[:, 3:6] * expand_times
    return large
