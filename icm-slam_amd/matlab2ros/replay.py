"""Recorded sequence -> the two ROS message streams, without a ROS network.

The reference ships a publisher (`scripts/matlab2ros/createbag.py:37-153`) that turns
`data_IJAC2018.mat` into `sensor_msgs/LaserScan` + `nav_msgs/Odometry` messages at 10 Hz over
rosbridge, for its online initialisation to subscribe to.  Sensor I/O is outside the accelerated
path, but the message layout is the interface of `Lidar.callback` / `Odometria.callback`
(`sensors_definitions.py`), so this module builds the same messages as plain dicts
(roslibpy's wire format) and hands them straight to the callbacks:

    lidar, odo = Lidar(config=cfg), Odometria(config=cfg)
    replay(observations, odometry, velocities, lidar.callback, odo.callback)
    ICM.load_messages(lidar, odo)          # -> mediciones / odometria / u, as load_data() does

Message fields follow the ROS definitions; the scan geometry (181 beams from -90 to +90
degrees, 1 degree apart) and the 0.1 s sample period are the dataset's.
"""
import math

import numpy as np

SAMPLE_PERIOD = 0.1          # [s] createbag.py publishes at 10 Hz
# The 36-element covariance the reference's publisher sends with every pose and twist (createbag.py:86-88,99-101): its
# non-zero entries sit at flat indices 0, 6, 12 (0.001) and 18, 24, 30 (100.0) -- every sixth element, NOT the diagonal
# 0, 7, 14, ... of a row-major 6x6 matrix.  Reproduced as sent (pinned by tests/golden/createbag_messages.json).
_COV = [0.001 if i in (0, 6, 12) else (100.0 if i in (18, 24, 30) else 0.0) for i in range(36)]


def stamp_of(seq, period=SAMPLE_PERIOD):
    s = seq * period
    return {"secs": int(s), "nsecs": int((s - int(s)) * 10 ** 9)}


def laser_scan_message(ranges, seq, period=SAMPLE_PERIOD):
    """sensor_msgs/LaserScan of one scan column (B beams, first beam at -90 degrees)."""
    return {
        "header": {"seq": int(seq), "stamp": stamp_of(seq, period), "frame_id": "Lidar_horizontal"},
        "angle_min": -math.pi / 2, "angle_max": math.pi / 2, "angle_increment": math.pi / 180,
        "time_increment": 0.0, "scan_time": 0.0, "range_min": 0.5, "range_max": 20.0,
        "ranges": [float(r) for r in np.asarray(ranges).reshape(-1)], "intensities": [],
    }


def odometry_message(pose, velocity, seq, period=SAMPLE_PERIOD):
    """nav_msgs/Odometry of one sample: planar pose (x, y, yaw) as position + quaternion about z,
    twist (v, w) as linear.x / angular.z."""
    yaw = float(pose[2])
    return {
        "header": {"seq": int(seq), "stamp": stamp_of(seq, period), "frame_id": "odom_groundtruth"},
        "child_frame_id": "base_link",
        "pose": {"pose": {"position": {"x": float(pose[0]), "y": float(pose[1]), "z": 0.0},
                          "orientation": {"x": 0.0, "y": 0.0, "z": math.sin(yaw / 2), "w": math.cos(yaw / 2)}},
                 "covariance": list(_COV)},
        "twist": {"twist": {"linear": {"x": float(velocity[0]), "y": 0.0, "z": 0.0},
                            "angular": {"x": 0.0, "y": 0.0, "z": float(velocity[1])}},
                  "covariance": list(_COV)},
    }


def messages(observations, odometry, velocities, period=SAMPLE_PERIOD):
    """Generator of (laser_scan, odometry) message pairs, one per sample, sequence numbers from 0."""
    observations = np.asarray(observations)
    for k in range(observations.shape[1]):
        yield (laser_scan_message(observations[:, k], k, period),
               odometry_message(odometry[:, k], velocities[:, k], k, period))


def replay(observations, odometry, velocities, on_scan, on_odometry, period=SAMPLE_PERIOD):
    """Feed every sample to the two subscriber callbacks in publication order; returns the count."""
    n = 0
    for scan, odo in messages(observations, odometry, velocities, period):
        on_odometry(odo)
        on_scan(scan)
        n += 1
    return n
