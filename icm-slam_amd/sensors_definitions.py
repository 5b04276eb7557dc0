"""Counterpart of the reference's `scripts/sensors_definitions.py`: message parsers of the two
ROS topics.  Sensor I/O is outside the accelerated path; these exist so that code importing
`Lidar` / `Odometria` keeps importing, and so that recorded messages can be turned into the
arrays the sweep reads.  (The reference's `np.float`, removed from NumPy, is not used.)"""
import math

import numpy as np

from ICM_SLAM_tools import Sensor


def _stamp(msg):
    return msg['header']['stamp']['secs'] + msg['header']['stamp']['nsecs'] * 1e-9


class Lidar(Sensor):
    def __init__(self, **argd):
        Sensor.__init__(self, **argd)

    def callback(self, msg):
        """LaserScan -> one (B,1) column: NaN -> max range, + trunk radius, clip
        (reference scripts/sensors_definitions.py:20-22); scans that are not 180 beams are
        resampled to 1 degree starting at -90 degrees (:23-29)."""
        z = np.array([msg['ranges']], dtype=np.float64)
        z[np.isnan(z)] = self.config.rango_laser_max
        z = np.minimum(z + self.config.radio, z * 0.0 + self.config.rango_laser_max)
        if z.shape[1] != 180:
            s0 = int((-np.pi / 2 - msg['angle_min']) / msg['angle_increment'])
            step = round((np.pi / 180.0) / msg['angle_increment'])
            z = z[:, s0:step * 180:step]
        self.msgs.append({'seq': msg['header']['seq'], 'stamp': _stamp(msg), 'data': z.T})
        if callable(self.principalCallback):
            self.principalCallback()


class Odometria(Sensor):
    def __init__(self, **argd):
        Sensor.__init__(self, **argd)

    def callback(self, msg):
        """Odometry -> pose (x, y, yaw) and twist (v, w) (reference
        scripts/sensors_definitions.py:40-71)."""
        p = msg['pose']['pose']
        q = p['orientation']
        yaw = math.atan2(2.0 * (q['w'] * q['z'] + q['x'] * q['y']), 1.0 - 2.0 * (q['y'] ** 2 + q['z'] ** 2))
        odo = np.array([[p['position']['x'], p['position']['y'], yaw]]).T
        tw = msg['twist']['twist']
        vel = np.array([[tw['linear']['x'], tw['angular']['z']]]).T
        self.msgs.append({'seq': msg['header']['seq'], 'stamp': _stamp(msg), 'data': {'odo': odo, 'u': vel}})
        if callable(self.principalCallback):
            self.principalCallback()
