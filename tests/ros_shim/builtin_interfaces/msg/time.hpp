// TEST INFRASTRUCTURE — API-shaped stand-in for builtin_interfaces/msg/time.hpp (ROS 2 is not in the build image).
#pragma once
#include <cstdint>
namespace builtin_interfaces {
namespace msg {
struct Time {
  int32_t sec = 0;
  uint32_t nanosec = 0;
};
}  // namespace msg
}  // namespace builtin_interfaces
