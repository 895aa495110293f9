// TEST INFRASTRUCTURE — decomp_ros_msgs/msg/Polyhedron (geometry_msgs/Point[] points, geometry_msgs/Point[] normals) and
// PolyhedronArray (Header header, Polyhedron[] polyhedrons), decomp_ros/decomp_ros_msgs/msg/*.msg of the reference.
#pragma once
#include <geometry_msgs/msg/pose_stamped.hpp>
#include <memory>
#include <vector>
namespace decomp_ros_msgs {
namespace msg {
struct Polyhedron {
  std::vector<geometry_msgs::msg::Point> points, normals;
};
struct PolyhedronArray {
  std_msgs::msg::Header header;
  std::vector<Polyhedron> polyhedrons;
  using SharedPtr = std::shared_ptr<PolyhedronArray>;
};
}  // namespace msg
}  // namespace decomp_ros_msgs
