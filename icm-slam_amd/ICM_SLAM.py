"""The reference ships `scripts/ICM_SLAM.py` as a byte-identical copy of
`scripts/ICM_SLAM_tools.py`; same here: one module, two names."""
from ICM_SLAM_tools import *  # noqa: F401,F403
from ICM_SLAM_tools import ConfigICM, Mapa, ROS, Sensor  # noqa: F401
