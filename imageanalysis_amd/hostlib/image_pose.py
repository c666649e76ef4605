"""A minimal image record with the pose accessors the BA / matching modules call on the
reference's lib.image.Image objects (scripts/lib/image.py:462-521): property-tree backed
camera_pose / camera_pose_opt {ned[3], yaw_deg, pitch_deg, roll_deg, quat[4] (w first,
euler 'rzyx')}.  Used by tests and the synthetic BA generator; inside the reference tree
the real Image class is used."""
import numpy as np

from .._deps import getNode
from . import transforms as tf

d2r = np.pi / 180.0
r2d = 180.0 / np.pi


class PoseImage(object):
    def __init__(self, name):
        self.name = name
        self.node = getNode("/images/" + name, True)
        self.kp_list = []
        self.des_list = None
        self.match_list = {}
        self.matches_clean = True
        self.uv_list = []
        self.num_features = 0
        self.placed = False
        self.desc_timestamp = 0.0
        self.cam2body = np.array([[0, 0, 1], [1, 0, 0], [0, 1, 0]], dtype=float)   # image.py:50-54
        self.body2cam = np.linalg.inv(self.cam2body)

    def set_camera_pose(self, ned, yaw_deg, pitch_deg, roll_deg, opt=False):
        quat = tf.quaternion_from_euler(yaw_deg * d2r, pitch_deg * d2r, roll_deg * d2r, 'rzyx')
        if opt:
            node = self.node.getChild('camera_pose_opt', True)
            node.setBool('valid', True)
        else:
            node = self.node.getChild('camera_pose', True)
            ac = self.node.getChild('aircraft_pose', False) if self.node.hasChild('aircraft_pose') else None
            if ac is not None and ac.hasChild('derived_from_camera_pose'):
                # (an aircraft pose this stand-in derived from an EARLIER camera pose of the node)
                del self.node.__dict__['aircraft_pose']
        for i in range(3):
            node.setFloatEnum('ned', i, ned[i])
        node.setFloat('yaw_deg', yaw_deg)
        node.setFloat('pitch_deg', pitch_deg)
        node.setFloat('roll_deg', roll_deg)
        node.setLen('quat', 4)
        for i in range(4):
            node.setFloatEnum('quat', i, quat[i])

    def get_camera_pose(self, opt=False):
        node = self.node.getChild('camera_pose_opt' if opt else 'camera_pose', True)
        ned = [node.getFloatEnum('ned', i) for i in range(3)]
        ypr = [node.getFloat('yaw_deg'), node.getFloat('pitch_deg'), node.getFloat('roll_deg')]
        quat = [node.getFloatEnum('quat', i) for i in range(4)]
        return ned, ypr, quat

    def get_body2ned(self, opt=False):
        ned, ypr, quat = self.get_camera_pose(opt)
        return tf.quaternion_matrix(np.array(quat))[:3, :3]

    def set_aircraft_pose(self, lat_deg, lon_deg, alt_m, yaw_deg, pitch_deg, roll_deg,
                          flight_time=-1.0):
        """scripts/lib/image.py:415-432 (the values find_matches' yaw estimate reads back)"""
        node = self.node.getChild('aircraft_pose', True)
        for k, v in (('lat_deg', lat_deg), ('lon_deg', lon_deg), ('alt_m', alt_m),
                     ('yaw_deg', yaw_deg), ('pitch_deg', pitch_deg), ('roll_deg', roll_deg)):
            node.setFloat(k, v)
        quat = tf.quaternion_from_euler(yaw_deg * d2r, pitch_deg * d2r, roll_deg * d2r, 'rzyx')
        node.setLen('quat', 4)
        for i in range(4):
            node.setFloatEnum('quat', i, quat[i])

    def get_aircraft_pose(self):
        node = self.node.getChild('aircraft_pose', True)
        lla = [node.getFloat('lat_deg'), node.getFloat('lon_deg'), node.getFloat('alt_m')]
        ypr = [node.getFloat('yaw_deg'), node.getFloat('pitch_deg'), node.getFloat('roll_deg')]
        quat = [node.getFloatEnum('quat', i) for i in range(4)]
        return lla, ypr, quat

    def set_aircraft_yaw_error_estimate(self, yaw_error_deg):
        """scripts/lib/image.py:434-457: the aircraft quaternion with the yaw bias added, and the
        CAMERA pose that follows from it (mount offset applied) -- what the next pair of
        find_matches triangulates with"""
        from . import camera
        ac = self.node.getChild('aircraft_pose', True)
        if not ac.hasChild('yaw_deg'):
            # (stand-in only: a synthetic project that logged CAMERA poses and never an aircraft
            #  pose -- the reference always has one, lib/pose.py:125-152 derives the camera pose
            #  from it -- gets the aircraft attitude that leads to its camera pose under the
            #  configured mount, so that the re-derivation below lands on that pose + the yaw error)
            ned, ypr, _q = self.get_camera_pose()
            self.set_pose_from_camera(ned, *ypr)
        ac.setFloat("yaw_error_deg", yaw_error_deg)
        yaw_deg, pitch_deg, roll_deg = ac.getFloat('yaw_deg'), ac.getFloat('pitch_deg'), ac.getFloat('roll_deg')
        ned2body = tf.quaternion_from_euler((yaw_deg + yaw_error_deg) * d2r, pitch_deg * d2r,
                                            roll_deg * d2r, 'rzyx')
        ac.setLen('quat', 4)
        for i in range(4):
            ac.setFloatEnum('quat', i, ned2body[i])
        ned2cam = tf.quaternion_multiply(ned2body, camera.get_body2cam())
        yaw_rad, pitch_rad, roll_rad = tf.euler_from_quaternion(ned2cam, 'rzyx')
        cp = self.node.getChild('camera_pose', True)
        cp.setFloat('yaw_deg', yaw_rad * r2d)
        cp.setFloat('pitch_deg', pitch_rad * r2d)
        cp.setFloat('roll_deg', roll_rad * r2d)
        cp.setLen('quat', 4)
        for i in range(4):
            cp.setFloatEnum('quat', i, ned2cam[i])

    def set_pose_from_camera(self, ned, yaw_deg, pitch_deg, roll_deg, lla=(45.0, -93.0, 300.0)):
        """Synthetic projects (tests, bench) log a CAMERA pose; the reference derives the camera
        pose from the AIRCRAFT pose and the mount offset (lib/pose.py:125-152) and re-derives it
        whenever find_matches updates an image's yaw-error estimate (lib/image.py:434-457).  This
        stores the camera pose and the aircraft attitude that leads to it under the configured
        mount, so that the re-derivation lands on the same pose (plus the estimated yaw error)."""
        from . import camera
        self.set_camera_pose(ned, yaw_deg, pitch_deg, roll_deg)
        ned2cam = tf.quaternion_from_euler(yaw_deg * d2r, pitch_deg * d2r, roll_deg * d2r, 'rzyx')
        b = camera.get_body2cam()
        inv = np.array([b[0], -b[1], -b[2], -b[3]]) / float(np.dot(b, b))
        y, p, r = tf.euler_from_quaternion(tf.quaternion_multiply(ned2cam, inv), 'rzyx')
        self.set_aircraft_pose(lla[0], lla[1], lla[2], y * r2d, p * r2d, r * r2d)
        self.node.getChild('aircraft_pose', True).setBool('derived_from_camera_pose', True)

    def get_cam2body(self):
        return self.cam2body

    def get_body2cam(self):
        return self.body2cam

    def detect_features(self, scale, use_cache=True):
        raise RuntimeError("PoseImage carries no pixels: attach kp_list/des_list yourself")

    def save_matches(self):
        self.matches_clean = True


class PoseProject(object):
    """The slice of lib.project.ProjectMgr the hot path touches."""

    def __init__(self, names, analysis_dir=None):
        self.analysis_dir = analysis_dir
        self.image_list = [PoseImage(n) for n in names]

    def findIndexByName(self, name):
        for i, im in enumerate(self.image_list):
            if im.name == name:
                return i
        return None

    def findImageByName(self, name):
        for im in self.image_list:
            if im.name == name:
                return im
        return None

    def save_images_info(self):
        pass
