// hdsm_agent_node.cpp — next row f3: an rclcpp node with the wire surface of multi_agent_planner::Agent's trajectory exchange,
// driving the C ABI (include/hdsm.h, hdsm_swarm.h, hdsm_stats.h) for a SHARD of agents hosted by one process/GPU.
// AC = multi_agent_planner/src/agent_class.cpp of lis-epfl/multi_agent_pkgs.
//
// Compiled only where ROS 2 and the reference's message package are installed (neither is in the build image of this
// repository, so here the file compiles to a stub main that says so — see ros/README.md). What it mirrors:
//   parameters   n_rob, id (first agent of the shard), n_local, n_hor, dt, step_plan, topic_name, state_ini / goal per agent,
//                save_stats (AC:2190-2308, the subset this path needs)
//   publishes    <topic_name>_<id>/traj_full  (multi_agent_planner_msgs/Trajectory, AC:46-48, filled as in AC:645-677)
//   subscribes   <topic_name>_<k>/traj_full for every agent k NOT hosted here (AC:610-627), stored like
//                TrajectoryOtherAgentsCallback (AC:629-643), latency recorded (com_latency_ms_)
//   loop         a wall timer of dt * step_plan whose callback is one lock-step round (AC:157-258): prepare -> hdsm_replan ->
//                commit -> publish; agents hosted here exchange their plans in memory, the others through DDS
//   shutdown     rclcpp::on_shutdown -> hdsm_swarm_shutdown = Agent::OnShutdown (AC:2446-2466)
#if __has_include(<rclcpp/rclcpp.hpp>) && __has_include(<multi_agent_planner_msgs/msg/trajectory.hpp>)
#include <chrono>
#include <multi_agent_planner_msgs/msg/trajectory.hpp>
#include <mutex>
#include <rclcpp/rclcpp.hpp>
#include <string>
#include <vector>

#include "../include/hdsm.h"
#include "../include/hdsm_stats.h"
#include "../include/hdsm_swarm.h"

using Trajectory = multi_agent_planner_msgs::msg::Trajectory;

class HdsmAgents : public rclcpp::Node {
 public:
  HdsmAgents() : Node("hdsm_agent_node") {
    n_rob_ = declare_parameter("n_rob", 1);
    first_ = declare_parameter("id", 0);
    n_local_ = declare_parameter("n_local", 1);
    n_hor_ = declare_parameter("n_hor", 9);
    save_stats_ = declare_parameter("save_stats", false);
    topic_ = declare_parameter("topic_name", std::string("agent"));
    const std::vector<double> ini = declare_parameter("state_ini", std::vector<double>(3 * n_local_, 0.0));  // [n_local][3]
    const std::vector<double> goal = declare_parameter("goal", std::vector<double>(3 * n_local_, 0.0));
    hdsm_default_params(&prm_, n_hor_);
    prm_.dt = declare_parameter("dt", 0.1);
    hdsm_swarm_default_config(&cfg_);
    cfg_.step_plan = declare_parameter("step_plan", 1);
    if (hdsm_create(&prm_, n_local_, n_rob_, declare_parameter("device", 0), &solver_) != HDSM_OK)
      throw std::runtime_error(std::string("hdsm_create: ") + hdsm_last_error());
    if (hdsm_swarm_create(&prm_, &cfg_, n_rob_, first_, n_local_, ini.data(), goal.data(), &swarm_) != HDSM_OK)
      throw std::runtime_error("hdsm_swarm_create");
    const int N = n_hor_, P = prm_.poly_hor, RS = prm_.max_rows_static;
    plans_.assign((size_t)n_rob_ * (N + 1) * 9, 0.0), has_.assign(n_rob_, 0);
    id_.resize(n_local_), n_poly_.resize(n_local_), n_rows_.resize(n_local_ * P), status_.resize(n_local_);
    state_.resize(9 * n_local_), ref_.resize(6 * N * n_local_), A_.resize((size_t)n_local_ * P * RS * 3), b_.resize((size_t)n_local_ * P * RS);
    traj_.resize((size_t)n_local_ * (N + 1) * 9), ctrl_.resize((size_t)n_local_ * N * 3), obj_.resize(n_local_), used_.resize(n_local_ * P);
    local_.resize(traj_.size()), has_local_.resize(n_local_);
    stats_remote_.resize(n_rob_, nullptr);
    for (int k = 0; k < n_local_; ++k)
      pubs_.push_back(create_publisher<Trajectory>(topic_ + "_" + std::to_string(first_ + k) + "/traj_full", 10));
    for (int k = 0; k < n_rob_; ++k) {  // CreateTrajectorySubsriberVector, AC:610-627
      if (k >= first_ && k < first_ + n_local_) continue;
      subs_.push_back(create_subscription<Trajectory>(topic_ + "_" + std::to_string(k) + "/traj_full", 10,
                                                      [this, k](const Trajectory::SharedPtr msg) { on_other(*msg, k); }));
    }
    timer_ = create_wall_timer(std::chrono::duration<double>(prm_.dt * cfg_.step_plan), [this] { round(); });
    rclcpp::on_shutdown([this] { shutdown(); });
  }
  // introspection (tests): agents hosted elsewhere whose plan has arrived over traj_full; lock-step rounds done
  int remote_plans_known() {
    std::lock_guard<std::mutex> g(mtx_);
    int cnt = 0;
    for (int k = 0; k < n_rob_; ++k) cnt += (k < first_ || k >= first_ + n_local_) && has_[k];
    return cnt;
  }
  int rounds() const { return rounds_; }

 private:
  void on_other(const Trajectory& msg, int k) {  // TrajectoryOtherAgentsCallback, AC:629-643
    std::lock_guard<std::mutex> g(mtx_);
    const int N = n_hor_;
    if ((int)msg.states.size() < N + 1) return;
    for (int i = 0; i <= N; ++i)
      for (int c = 0; c < 3; ++c) {
        double* rec = &plans_[((size_t)k * (N + 1) + i) * 9];
        rec[c] = msg.states[i].position[c], rec[3 + c] = msg.states[i].velocity[c];
        rec[6 + c] = msg.states[i].acceleration.size() > (size_t)c ? msg.states[i].acceleration[c] : 0.0;
      }
    has_[k] = 1;
    latency_.push_back({k, (now() - rclcpp::Time(msg.stamp)).seconds() * 1e3});
  }

  void round() {  // one iteration of TrajPlanningIteration (AC:157-258) for the whole shard
    const int N = n_hor_;
    std::vector<double> plans;
    std::vector<uint8_t> has;
    {
      std::lock_guard<std::mutex> g(mtx_);
      plans = plans_, has = has_;
    }
    hdsm_swarm_prepare(swarm_, plans.data(), has.data(), id_.data(), state_.data(), ref_.data(), n_poly_.data(), n_rows_.data(),
                       A_.data(), b_.data());
    const auto t0 = std::chrono::steady_clock::now();
    hdsm_replan(solver_, n_local_, n_rob_, id_.data(), state_.data(), ref_.data(), n_poly_.data(), n_rows_.data(), A_.data(), b_.data(),
                plans.data(), has.data(), traj_.data(), ctrl_.data(), used_.data(), status_.data(), obj_.data());
    hdsm_swarm_record_solve_ms(swarm_, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
    hdsm_swarm_commit(swarm_, traj_.data(), ctrl_.data(), used_.data(), status_.data(), local_.data(), has_local_.data());
    std::lock_guard<std::mutex> g(mtx_);
    for (int k = 0; k < n_local_; ++k) {
      if (!has_local_[k]) continue;
      std::copy_n(&local_[(size_t)k * (N + 1) * 9], (N + 1) * 9, &plans_[(size_t)(first_ + k) * (N + 1) * 9]);
      has_[first_ + k] = 1;
      Trajectory msg;  // PublishTrajectoryFull, AC:645-677
      msg.stamp = now();
      msg.yaw = 0.0;
      msg.dt = prm_.dt;
      for (int i = 0; i <= N; ++i) {
        const double* rec = &local_[((size_t)k * (N + 1) + i) * 9];
        multi_agent_planner_msgs::msg::State st;
        st.position = {rec[0], rec[1], rec[2]}, st.velocity = {rec[3], rec[4], rec[5]}, st.acceleration = {rec[6], rec[7], rec[8]};
        msg.states.push_back(st);
      }
      pubs_[k]->publish(msg);
    }
    ++rounds_;
  }

  void shutdown() {  // Agent::OnShutdown, AC:2446-2466, for every hosted agent
    std::vector<char> report(1 << 16);
    for (int k = 0; k < n_local_; ++k)
      if (hdsm_swarm_shutdown(swarm_, k, ".", save_stats_ ? 1 : 0, report.data(), (int)report.size()) > 0)
        RCLCPP_INFO(get_logger(), "%s", report.data());
  }

  struct Lat {
    int from;
    double ms;
  };
  hdsm_params prm_;
  hdsm_swarm_config cfg_;
  void *solver_ = nullptr, *swarm_ = nullptr;
  int n_rob_, first_, n_local_, n_hor_, rounds_ = 0;
  bool save_stats_;
  std::string topic_;
  std::mutex mtx_;
  std::vector<double> plans_, state_, ref_, A_, b_, traj_, ctrl_, obj_, local_;
  std::vector<uint8_t> has_, used_, has_local_;
  std::vector<int32_t> id_, n_poly_, n_rows_, status_;
  std::vector<void*> stats_remote_;
  std::vector<Lat> latency_;
  std::vector<rclcpp::Publisher<Trajectory>::SharedPtr> pubs_;
  std::vector<rclcpp::Subscription<Trajectory>::SharedPtr> subs_;
  rclcpp::TimerBase::SharedPtr timer_;
};

int main(int argc, char** argv) {
  rclcpp::init(argc, argv);
  rclcpp::spin(std::make_shared<HdsmAgents>());
  rclcpp::shutdown();
  return 0;
}
#else
#include <cstdio>
int main() {
  std::puts("hdsm_agent_node: built without ROS 2 (rclcpp / multi_agent_planner_msgs headers not found); the node wrapper is "
            "compiled out. The ROS-free half of row f3 (timing records, shutdown CSVs) lives in libhdsm.so: include/hdsm_stats.h.");
  return 0;
}
#endif
