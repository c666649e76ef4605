"""Camera calibration accessors on /config/camera -- same names and meaning as the
reference's scripts/lib/camera.py:58-139 (get_K, get_dist_coeffs, get_image_params, the mount
angles and get_body2cam + setters)."""
import numpy as np

from .._deps import getNode
from . import transforms as tf

camera_node = getNode('/config/camera', True)


def get_K(optimized=False):
    """3x3 K from the 9-vector 'K' (or 'K_opt' when optimized and present) -- camera.py:58-75."""
    key = 'K_opt' if optimized and camera_node.hasChild('K_opt') else 'K'
    return np.array([camera_node.getFloatEnum(key, i) for i in range(9)]).reshape(3, 3)


def set_K(fx, fy, cu, cv, optimized=False):
    key = 'K_opt' if optimized else 'K'
    vals = [fx, 0.0, cu, 0.0, fy, cv, 0.0, 0.0, 1.0]
    camera_node.setLen(key, 9)
    for i, v in enumerate(vals):
        camera_node.setFloatEnum(key, i, v)


def get_dist_coeffs(optimized=False):
    """[k1, k2, p1, p2, k3] -- camera.py:94-103."""
    key = 'dist_coeffs_opt' if optimized and camera_node.hasChild('dist_coeffs_opt') \
        else 'dist_coeffs'
    return np.array([camera_node.getFloatEnum(key, i) for i in range(5)])


def set_dist_coeffs(dist_coeffs, optimized=False):
    key = 'dist_coeffs_opt' if optimized else 'dist_coeffs'
    camera_node.setLen(key, 5)
    for i in range(5):
        camera_node.setFloatEnum(key, i, dist_coeffs[i])


def set_image_params(width_px, height_px):
    camera_node.setInt('width_px', width_px)
    camera_node.setInt('height_px', height_px)


def get_image_params():
    return camera_node.getInt('width_px'), camera_node.getInt('height_px')


def set_mount_params(yaw_deg, pitch_deg, roll_deg):
    """camera.py:123-127"""
    mount_node = camera_node.getChild('mount', True)
    mount_node.setFloat('yaw_deg', yaw_deg)
    mount_node.setFloat('pitch_deg', pitch_deg)
    mount_node.setFloat('roll_deg', roll_deg)


def get_mount_params():
    mount_node = camera_node.getChild('mount', True)
    return [mount_node.getFloat('yaw_deg'), mount_node.getFloat('pitch_deg'),
            mount_node.getFloat('roll_deg')]


def get_body2cam():
    """the mount offset as a (w,x,y,z) quaternion -- camera.py:136-139"""
    d2r = np.pi / 180.0
    yaw_deg, pitch_deg, roll_deg = get_mount_params()
    return tf.quaternion_from_euler(yaw_deg * d2r, pitch_deg * d2r, roll_deg * d2r, 'rzyx')
