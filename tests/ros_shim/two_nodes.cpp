// TEST INFRASTRUCTURE — two hdsm_agent_node nodes in ONE process on the in-memory bus of tests/ros_shim/rclcpp/rclcpp.hpp:
// each hosts half of a small circular exchange and learns the other half's plans ONLY through traj_full messages
// (publishers AC:46-48 / 645-677, subscriptions AC:610-643). Prints counters the test checks. usage: two_nodes [ticks]
#define main hdsm_agent_node_main
#include "../../ros/hdsm_agent_node.cpp"
#undef main
#include <cmath>
#include <map>

int main(int argc, char** argv) {
  const int ticks = argc > 1 ? std::atoi(argv[1]) : 12;
  const int n_rob = 8, per = 4;
  rclcpp::init(argc, argv);
  std::vector<std::shared_ptr<HdsmAgents>> nodes;
  try {
    for (int r = 0; r < 2; ++r) {
      std::vector<double> ini, goal;
      for (int k = 0; k < per; ++k) {
        const int a = r * per + k, g = (a + n_rob / 2) % n_rob;
        const double R = 6.0, pi = 3.14159265358979323846;
        ini.insert(ini.end(), {18 + R * std::cos(2 * pi * a / n_rob), 15 + R * std::sin(2 * pi * a / n_rob), 1.5});
        goal.insert(goal.end(), {18 + R * std::cos(2 * pi * g / n_rob), 15 + R * std::sin(2 * pi * g / n_rob), 1.5});
      }
      auto& ov = rclcpp::shim::bus().overrides;
      ov["n_rob"] = n_rob, ov["id"] = r * per, ov["n_local"] = per, ov["n_hor"] = 10, ov["state_ini"] = ini, ov["goal"] = goal;
      nodes.push_back(std::make_shared<HdsmAgents>());
    }
  } catch (const std::exception& e) {
    std::printf("node construction failed: %s\n", e.what());
    return 7;  // (on a box without a GPU: hdsm_create has no CPU fallback)
  }
  for (int k = 0; k < ticks; ++k) rclcpp::spin_some_all();
  long pub = 0, del = 0;
  for (auto& kv : rclcpp::shim::bus().published) pub += kv.second;
  for (auto& kv : rclcpp::shim::bus().delivered) del += kv.second;
  std::printf("topics %zu published %ld delivered %ld\n", rclcpp::shim::bus().published.size(), pub, del);
  for (int r = 0; r < 2; ++r) std::printf("node %d: remote plans known %d of %d, rounds %d\n", r, nodes[r]->remote_plans_known(), n_rob - per, nodes[r]->rounds());
  // per topic kind: messages published (8 kinds x 8 agents x rounds, minus the first rounds' missing trajectories)
  std::map<std::string, long> kinds;
  for (auto& kv : rclcpp::shim::bus().published) kinds[kv.first.substr(kv.first.rfind('/') + 1)] += kv.second;
  for (auto& kv : kinds) std::printf("kind %s %ld\n", kv.first.c_str(), kv.second);
  // agent 0 starts at angle 0 of the circle and flies towards -x: its yaw turns towards pi (ComputeYawAngle)
  std::printf("yaw agent0 %.6f agent2 %.6f\n", nodes[0]->yaw(0), nodes[0]->yaw(2));
  rclcpp::shutdown();
  return 0;
}
