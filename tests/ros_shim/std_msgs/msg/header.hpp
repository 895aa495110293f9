// TEST INFRASTRUCTURE — the C++ struct rosidl generates for std_msgs/msg/Header (stamp, frame_id), as far as the node uses it.
#pragma once
#include <builtin_interfaces/msg/time.hpp>
#include <string>
namespace std_msgs {
namespace msg {
struct Header {
  builtin_interfaces::msg::Time stamp;
  std::string frame_id;
};
}  // namespace msg
}  // namespace std_msgs
