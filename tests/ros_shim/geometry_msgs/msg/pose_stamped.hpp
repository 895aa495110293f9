// TEST INFRASTRUCTURE — geometry_msgs/msg/{Point, Point32, Quaternion, Vector3, Pose, PoseStamped}: the fields of the ROS 2 definitions.
#pragma once
#include <std_msgs/msg/header.hpp>
namespace geometry_msgs {
namespace msg {
struct Point {
  double x = 0, y = 0, z = 0;
};
struct Point32 {
  float x = 0, y = 0, z = 0;
};
struct Vector3 {
  double x = 0, y = 0, z = 0;
};
struct Quaternion {
  double x = 0, y = 0, z = 0, w = 1;
};
struct Pose {
  Point position;
  Quaternion orientation;
};
struct PoseStamped {
  std_msgs::msg::Header header;
  Pose pose;
};
}  // namespace msg
}  // namespace geometry_msgs
