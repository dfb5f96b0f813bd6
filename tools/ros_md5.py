#!/usr/bin/env python3
"""ROS-1 md5sums of the nine vision_slam_frontend messages (MESSAGES below: the field lists of the reference's msg/*.msg, the
wire contract host/slam_to_ros.h serialises), by genmsg's rule (gentools.compute_md5_text, what `rosmsg md5` / gendeps print):

    comments and blank lines dropped; constants first as "type name=value"; then one line per field, "type name" for a
    builtin type (array suffix kept) and "<md5 of the sub-message> name" for a message type (array suffix dropped);
    lines joined by '\\n', no trailing newline; md5 of that text.

A bag writer needs them for its connection headers (md5sum=, type=vision_slam_frontend/<Name>).  geometry_msgs' three
messages are restated here; their md5sums are known constants of every ROS-1 distribution and pin the rule:
Point = Vector3 = 4a842b65f413084dc2b10fb484ea7f17, Quaternion = a779879fadf0160734f906b8c19c7004.

    python tools/ros_md5.py            prints the table
    python tools/ros_md5.py --header   prints the constants of host/slam_to_ros.h"""
import hashlib
import sys

MESSAGES = {  # message -> its field list, "type name" per line (comments of the .msg files do not enter the md5sum)
    'CameraExtrinsics': 'float64[3] translation\nfloat64[3] rotation',
    'CameraIntrinsics': 'float64 fx\nfloat64 fy\nfloat64 cx\nfloat64 cy',
    'FeatureMatch': 'uint64 id_initial\nuint64 id_current',
    'OdometryFactor': 'uint64 pose_i\nuint64 pose_j\ngeometry_msgs/Vector3 translation\ngeometry_msgs/Quaternion rotation',
    'RobotPose': 'geometry_msgs/Vector3 loc\ngeometry_msgs/Quaternion angle',
    'SLAMNode': 'uint64 id\nfloat64 timestamp\nRobotPose pose\nVisionFeature[] features',
    'SLAMProblem': 'SLAMNode[] nodes\nVisionFactor[] vision_factors\nOdometryFactor[] odometry_factors',
    'VisionFactor': 'uint64 pose_initial\nuint64 pose_current\nFeatureMatch[] feature_matches',
    'VisionFeature': 'uint64 id\ngeometry_msgs/Point pixel\ngeometry_msgs/Point point3d',
}
PACKAGE = "vision_slam_frontend"
BUILTIN = {"bool", "int8", "uint8", "int16", "uint16", "int32", "uint32", "int64", "uint64", "float32", "float64", "string",
           "time", "duration", "char", "byte"}
GEOMETRY = {
    "geometry_msgs/Point": "float64 x\nfloat64 y\nfloat64 z\n",
    "geometry_msgs/Vector3": "float64 x\nfloat64 y\nfloat64 z\n",
    "geometry_msgs/Quaternion": "float64 x\nfloat64 y\nfloat64 z\nfloat64 w\n",
}
KNOWN = {"geometry_msgs/Point": "4a842b65f413084dc2b10fb484ea7f17", "geometry_msgs/Vector3": "4a842b65f413084dc2b10fb484ea7f17",
         "geometry_msgs/Quaternion": "a779879fadf0160734f906b8c19c7004"}
NAMES = ["CameraExtrinsics", "CameraIntrinsics", "FeatureMatch", "OdometryFactor", "RobotPose", "SLAMNode", "SLAMProblem",
         "VisionFactor", "VisionFeature"]


def msg_text(full_name: str) -> str:
    if full_name in GEOMETRY:
        return GEOMETRY[full_name]
    pkg, name = full_name.split("/")
    assert pkg == PACKAGE, full_name
    return MESSAGES[name] + "\n"


def md5_text(full_name: str) -> str:
    pkg = full_name.split("/")[0]
    consts, fields = [], []
    for line in msg_text(full_name).splitlines():
        line = line.split("#", 1)[0].strip()
        if not line:
            continue
        type_, rest = line.split(None, 1)
        if "=" in rest:
            name, val = rest.split("=", 1)
            consts.append("%s %s=%s" % (type_, name.strip(), val.strip()))
            continue
        name = rest.strip()
        bare = type_.split("[", 1)[0]
        if bare in BUILTIN:
            fields.append("%s %s" % (type_, name))
        else:
            sub = bare if "/" in bare else ("std_msgs/Header" if bare == "Header" else "%s/%s" % (pkg, bare))
            fields.append("%s %s" % (md5(sub), name))
    return "\n".join(consts + fields)


def md5(full_name: str) -> str:
    return hashlib.md5(md5_text(full_name).encode()).hexdigest()


def table() -> dict:
    for k, v in KNOWN.items():
        assert md5(k) == v, (k, md5(k), v)  # the rule reproduces the md5sums every ROS-1 installation carries
    return {n: md5("%s/%s" % (PACKAGE, n)) for n in NAMES}


if __name__ == "__main__":
    t = table()
    if "--header" in sys.argv[1:]:
        for n in NAMES:
            print('constexpr const char* kMd5%s = "%s";' % (n, t[n]))
    else:
        for n in NAMES:
            print("%s  %s/%s" % (t[n], PACKAGE, n))
