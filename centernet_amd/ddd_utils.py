"""3-D box geometry of the ddd task -- host helpers behind ``ddd_post_process`` (mirror of the
numeric functions of src/lib/utils/ddd_utils.py:7-121; ``draw_box_3d`` belongs to the debugger and
is not built).  KITTI camera convention: x right, y down, z forward; a box is (h, w, l) with its
``location`` at the centre of the bottom face; ``P`` is the 3 x 4 projection matrix.

Every function keeps the reference's arithmetic types (float32 inputs stay float32, the rotation
matrix and the corner table are float32), so results are bit-identical to the reference's under
the same NumPy (tests/golden/gen_golden_tasks.py)."""
import numpy as np


def _rot_y(rotation_y):
    c, s = np.cos(rotation_y), np.sin(rotation_y)
    return np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]], dtype=np.float32)


def compute_box_3d(dim, location, rotation_y):
    """(h, w, l), bottom-centre location, yaw -> the 8 corners, (8, 3) float32 (ddd_utils.py:8-24):
    four of the bottom face (y = 0) then four of the top face (y = -h), l along x, w along z."""
    h, w, l = dim[0], dim[1], dim[2]
    half_l, half_w = l / 2, w / 2
    xs = [half_l, half_l, -half_l, -half_l] * 2
    ys = [0, 0, 0, 0, -h, -h, -h, -h]
    zs = [half_w, -half_w, -half_w, half_w] * 2
    corners = np.array([xs, ys, zs], dtype=np.float32)
    moved = np.dot(_rot_y(rotation_y), corners) + np.array(location, dtype=np.float32).reshape(3, 1)
    return moved.transpose(1, 0)


def project_to_image(pts_3d, P):
    """(n, 3) camera points -> (n, 2) pixels through the 3 x 4 matrix ``P`` (ddd_utils.py:26-35)."""
    homo = np.concatenate([pts_3d, np.ones((pts_3d.shape[0], 1), dtype=np.float32)], axis=1)
    uvw = np.dot(P, homo.transpose(1, 0)).transpose(1, 0)
    return uvw[:, :2] / uvw[:, 2:]


def compute_orientation_3d(dim, location, rotation_y):
    """The heading segment: box centre -> l ahead of it, (2, 3) float32 (ddd_utils.py:37-49)."""
    seg = np.array([[0, dim[2]], [0, 0], [0, 0]], dtype=np.float32)
    seg = np.dot(_rot_y(rotation_y), seg) + np.array(location, dtype=np.float32).reshape(3, 1)
    return seg.transpose(1, 0)


def unproject_2d_to_3d(pt_2d, depth, P):
    """Pixel + depth -> camera point, float32 (3,) (ddd_utils.py:68-78): the inverse of
    ``project_to_image`` for a matrix whose left 3 x 3 block is [[f, 0, cx], [0, f, cy], [0, 0, 1]]."""
    z = depth - P[2, 3]
    x = (pt_2d[0] * depth - P[0, 3] - P[0, 2] * z) / P[0, 0]
    y = (pt_2d[1] * depth - P[1, 3] - P[1, 2] * z) / P[1, 1]
    return np.array([x, y, z], dtype=np.float32)


def _wrap_pi(angle):
    if angle > np.pi:
        angle -= 2 * np.pi
    if angle < -np.pi:
        angle += 2 * np.pi
    return angle


def alpha2rot_y(alpha, x, cx, fx):
    """Observation angle -> yaw around the camera's y axis: the viewing ray's own angle
    atan2(x - cx, fx) is added back, result wrapped into [-pi, pi] (ddd_utils.py:80-92)."""
    return _wrap_pi(alpha + np.arctan2(x - cx, fx))


def rot_y2alpha(rot_y, x, cx, fx):
    """The inverse of ``alpha2rot_y`` (ddd_utils.py:94-106)."""
    return _wrap_pi(rot_y - np.arctan2(x - cx, fx))


def ddd2locrot(center, alpha, dim, depth, calib):
    """One detection: 2-D centre, observation angle, (h, w, l), depth -> (location of the bottom
    face centre, yaw) (ddd_utils.py:109-114).  The network's centre is the centre of the 3-D box,
    KITTI's location the bottom centre: y moves down by h / 2."""
    location = unproject_2d_to_3d(center, depth, calib)
    location[1] += dim[0] / 2
    return location, alpha2rot_y(alpha, center[0], calib[0, 2], calib[0, 0])


def project_3d_bbox(location, dim, rotation_y, calib):
    """The 8 corners in pixels, (8, 2) (ddd_utils.py:116-119)."""
    return project_to_image(compute_box_3d(dim, location, rotation_y), calib)
