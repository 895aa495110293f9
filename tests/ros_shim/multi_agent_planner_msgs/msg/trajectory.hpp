// TEST INFRASTRUCTURE — the C++ struct rosidl generates for multi_agent_planner_msgs/msg/Trajectory.msg:1-11
// (builtin_interfaces/Time stamp, float64 dt, State[] states, float64 yaw).
#pragma once
#include <builtin_interfaces/msg/time.hpp>
#include <memory>
#include <multi_agent_planner_msgs/msg/state.hpp>
#include <vector>
namespace multi_agent_planner_msgs {
namespace msg {
struct Trajectory {
  builtin_interfaces::msg::Time stamp;  // Trajectory.msg:2
  double dt = 0.0;                      // Trajectory.msg:5
  std::vector<State> states;            // Trajectory.msg:8
  double yaw = 0.0;                     // Trajectory.msg:11
  using SharedPtr = std::shared_ptr<Trajectory>;
};
}  // namespace msg
}  // namespace multi_agent_planner_msgs
