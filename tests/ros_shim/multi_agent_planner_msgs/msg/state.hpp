// TEST INFRASTRUCTURE — the C++ struct rosidl generates for multi_agent_planner_msgs/msg/State.msg:1-8 (three float64[] fields).
#pragma once
#include <memory>
#include <vector>
namespace multi_agent_planner_msgs {
namespace msg {
struct State {
  std::vector<double> position;      // State.msg:2
  std::vector<double> velocity;      // State.msg:5
  std::vector<double> acceleration;  // State.msg:8
  using SharedPtr = std::shared_ptr<State>;
};
}  // namespace msg
}  // namespace multi_agent_planner_msgs
