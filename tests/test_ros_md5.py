"""The md5sums host/slam_to_ros.h publishes for the nine vision_slam_frontend messages equal what genmsg's rule gives for the
committed field lists (tools/ros_md5.py: MESSAGES), and that rule reproduces the md5sums of geometry_msgs every ROS-1 installation
carries (tools/ros_md5.py)."""
import re
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "tools"))


def test_published_md5sums_follow_the_committed_field_lists():
    import ros_md5
    table = ros_md5.table()  # (asserts the known md5sums of geometry_msgs/Point, Vector3, Quaternion)
    header = (ROOT / "vision_slam_frontend_amd" / "host" / "slam_to_ros.h").read_text()
    found = dict(re.findall(r'kMd5(\w+) = "([0-9a-f]{32})"', header))
    assert found == table and len(table) == 9
    # the md5 text of a nested message carries the md5sums of its parts, arrays or not
    assert ros_md5.md5_text("vision_slam_frontend/SLAMNode") == "uint64 id\nfloat64 timestamp\n%s pose\n%s features" % (
        table["RobotPose"], table["VisionFeature"])
    assert ros_md5.md5_text("vision_slam_frontend/CameraExtrinsics") == "float64[3] translation\nfloat64[3] rotation"


def test_calibration_messages_of_the_default_config():
    """FrontendConfig::left_cam_to_robot carries the reference's literals (slam_frontend.cc:613-618) and the driver's two
    calibration messages (slam_frontend_main.cc:341-365) come out as ROS-1 payloads: CameraExtrinsics = translation + the
    rotation as scaled angle-axis (checked against the log map of the matrix in float64), CameraIntrinsics = fx fy cx cy."""
    import numpy as np
    sys.path.insert(0, str(ROOT))
    from vision_slam_frontend_amd import frontend
    fe = frontend.Frontend(0, 0, nfeatures=2000)  # (no image size: no GPU context is created)
    R, t = fe.left_cam_to_robot
    want_R = np.float32([[0.009916590468, -0.2835522866, 0.9589055021], [-0.9998698619, -0.01501486552, 0.005900269087],
                         [0.01272480238, -0.9588392225, -0.2836642819]])
    assert R.tobytes() == want_R.tobytes() and t.tobytes() == np.float32([-0.01, 0.06, 0.5299999713897705]).tobytes()
    ext, intr = fe.serialize_calibration()
    fe.close()
    assert len(ext) == 48 and len(intr) == 32
    e = np.frombuffer(ext, np.float64)
    assert e[:3].tobytes() == t.astype(np.float64).tobytes()
    Rd = want_R.astype(np.float64)
    angle = np.arccos((np.trace(Rd) - 1) / 2)
    axis = np.array([Rd[2, 1] - Rd[1, 2], Rd[0, 2] - Rd[2, 0], Rd[1, 0] - Rd[0, 1]])
    np.testing.assert_allclose(e[3:], axis / np.linalg.norm(axis) * angle, atol=2e-5)
    assert all(float(np.float32(v)) == v for v in e[3:])  # float values widened, as ExtrinsicsToRos assigns them
    k = np.frombuffer(intr, np.float64)
    assert k.tobytes() == np.float32([527.873518, 527.276819, 482.823413, 298.033945]).astype(np.float64).tobytes()
