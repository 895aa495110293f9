// TEST INFRASTRUCTURE — visualization_msgs/msg/Marker, the fields Agent::PublishCurrentPosition fills (AC:806-826).
#pragma once
#include <geometry_msgs/msg/pose_stamped.hpp>
#include <memory>
namespace std_msgs {
namespace msg {
struct ColorRGBA {
  float r = 0, g = 0, b = 0, a = 0;
};
}  // namespace msg
}  // namespace std_msgs
namespace visualization_msgs {
namespace msg {
struct Marker {
  static constexpr int32_t SPHERE = 2;
  static constexpr int32_t ADD = 0;
  std_msgs::msg::Header header;
  int32_t type = 0, action = 0;
  geometry_msgs::msg::Pose pose;
  geometry_msgs::msg::Vector3 scale;
  std_msgs::msg::ColorRGBA color;
  using SharedPtr = std::shared_ptr<Marker>;
};
}  // namespace msg
}  // namespace visualization_msgs
