"""MI355X-native batched EKF/UKF-SLAM predict–update engine (drop-in for the Filter::update path of
kevin-robb/live_ekf_slam).  Numerics live in the HIP extension libslam_hip.so behind include/slam_batch.h."""
from .config import SlamConfig, default_config, EKF_SLAM, UKF_LOC, UKF_SLAM, F64, F32  # noqa: F401
from .filters import BatchedEKF, BatchedUKF, BatchedUKFLoc, Command  # noqa: F401
from .pose_graph import BatchedPoseGraph, NaiveFilter  # noqa: F401
from ._lib import SlamError  # noqa: F401
