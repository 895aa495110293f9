// hdsm_agent_node.cpp — next row f3: an rclcpp node with the wire surface of multi_agent_planner::Agent's trajectory exchange,
// driving the C ABI (include/hdsm.h, hdsm_swarm.h, hdsm_stats.h) for a SHARD of agents hosted by one process/GPU.
// AC = multi_agent_planner/src/agent_class.cpp of lis-epfl/multi_agent_pkgs.
//
// Compiled only where ROS 2 and the reference's message package are installed (neither is in the build image of this
// repository, so here the file compiles to a stub main that says so — see ros/README.md). What it mirrors:
//   parameters   every parameter Agent::DeclareRosParameters / InitializeRosParameters knows (AC:2190-2308), with the reference's
//                defaults; those the hot path and its callers read go into hdsm_params / hdsm_swarm_config exactly as
//                InitializePlannerParameters does (AC:2169-2188), the others (mapping, path planner, Gurobi verbosity) are
//                accepted so that the reference's launch files start this node unchanged. Additions: n_local (agents hosted
//                here; state_ini / goal then hold n_local triples), device.
//   publishes    <topic_name>_<id>/traj_full  (multi_agent_planner_msgs/Trajectory, AC:46-48, filled as in AC:645-677, yaw from
//                ComputeYawAngle AC:1025-1051) and the rviz topics of AC:50-86: traj, traj_ref, path, traj_hist (nav_msgs/Path,
//                AC:679-780), polyhedra (decomp_ros_msgs/PolyhedronArray, AC:851-857), seeds (sensor_msgs/PointCloud2,
//                AC:829-849), position (visualization_msgs/Marker, AC:806-826)
//   subscribes   <topic_name>_<k>/traj_full for every agent k NOT hosted here (AC:610-627), stored like
//                TrajectoryOtherAgentsCallback (AC:629-643), latency recorded (com_latency_ms_)
//   loop         a wall timer of dt * step_plan whose callback is one lock-step round (AC:157-258): prepare -> hdsm_replan ->
//                commit -> publish; agents hosted here exchange their plans in memory, the others through DDS
//   shutdown     rclcpp::on_shutdown -> hdsm_swarm_shutdown = Agent::OnShutdown (AC:2446-2466)
#if __has_include(<rclcpp/rclcpp.hpp>) && __has_include(<multi_agent_planner_msgs/msg/trajectory.hpp>) && __has_include(<decomp_ros_msgs/msg/polyhedron_array.hpp>)
#include <chrono>
#include <cstring>
#include <decomp_ros_msgs/msg/polyhedron_array.hpp>
#include <multi_agent_planner_msgs/msg/trajectory.hpp>
#include <mutex>
#include <nav_msgs/msg/path.hpp>
#include <rclcpp/rclcpp.hpp>
#include <sensor_msgs/msg/point_cloud2.hpp>
#include <string>
#include <vector>
#include <visualization_msgs/msg/marker.hpp>

#include "../include/hdsm.h"
#include "../include/hdsm_stats.h"
#include "../include/hdsm_swarm.h"

using Trajectory = multi_agent_planner_msgs::msg::Trajectory;
using Path = nav_msgs::msg::Path;

class HdsmAgents : public rclcpp::Node {
 public:
  HdsmAgents() : Node("hdsm_agent_node") {
    // DeclareRosParameters / InitializeRosParameters, AC:2190-2308 (same names, same defaults)
    declare_parameter("get_grid_service_name", std::string("/env_builder_node/get_voxel_grid"));  // (mapping: out of scope, accepted)
    const std::vector<double> vg_range = declare_parameter("voxel_grid_range", std::vector<double>(3, 10.0));
    declare_parameter("publish_voxel_grid", false);
    declare_parameter("voxel_grid_update_period", 0.1);
    declare_parameter("use_mapping_util", true);
    declare_parameter("gurobi_verbose", true);
    topic_ = declare_parameter("topic_name", std::string("agent"));
    world_frame_ = declare_parameter("world_frame", std::string("world"));
    n_rob_ = declare_parameter("n_rob", 1);
    first_ = declare_parameter("id", 0);
    n_local_ = declare_parameter("n_local", 1);
    declare_parameter("n_x", 9);
    declare_parameter("n_u", 3);
    n_hor_ = declare_parameter("n_hor", 7);
    hdsm_default_params(&prm_, n_hor_);
    hdsm_swarm_default_config(&cfg_);
    prm_.dt = declare_parameter("dt", 0.1);
    cfg_.path_vel_min = declare_parameter("path_vel_min", 4.5);
    cfg_.path_vel_max = declare_parameter("path_vel_max", 4.5);
    cfg_.sens_dist = declare_parameter("sens_dist", 1.0);
    cfg_.sens_pot = declare_parameter("sens_pot", 1.0);
    cfg_.sens_other_agents = declare_parameter("sens_other_agents", 1.0);
    cfg_.path_vel_dec = declare_parameter("path_vel_dec", 0.1);
    declare_parameter("traj_ref_points_to_keep", 10);
    prm_.rk4 = declare_parameter("rk4", false) ? 1 : 0;
    cfg_.step_plan = declare_parameter("step_plan", 1);
    cfg_.thresh_dist = declare_parameter("thresh_dist", 0.2);
    prm_.poly_hor = declare_parameter("poly_hor", 3);
    cfg_.n_it_decomp = declare_parameter("n_it_decomp", 42);
    declare_parameter("use_cvx", true);
    cfg_.use_cvx_new = declare_parameter("use_cvx_new", false) ? 1 : 0;
    prm_.drone_radius = declare_parameter("drone_radius", 0.3);
    prm_.drone_z_offset = declare_parameter("drone_z_offset", 0.3);
    declare_parameter("path_infl_dist", 0.3);
    declare_parameter("com_latency", 0.0);
    prm_.r_u = declare_parameter("r_u", 0.01);
    const std::vector<double> r_x = declare_parameter("r_x", std::vector<double>(9, 0.0));
    const std::vector<double> r_n = declare_parameter("r_n", std::vector<double>(9, 0.0));
    for (int c = 0; c < 9; ++c) prm_.r_x[c] = c < (int)r_x.size() ? r_x[c] : 0.0, prm_.r_n[c] = c < (int)r_n.size() ? r_n[c] : 0.0;
    {  // InitializePlannerParameters, AC:2169-2188: velocity / acceleration boxes, jerk box (positions stay unbounded)
      const double max_vel = declare_parameter("max_vel", 9.5);
      const double max_acc_z = declare_parameter("max_acc_z", 30.0), min_acc_z = declare_parameter("min_acc_z", -30.0);
      const double max_acc_xy = declare_parameter("max_acc_xy", 30.0), min_acc_xy = declare_parameter("min_acc_xy", -30.0);
      const double max_jerk = declare_parameter("max_jerk", 60.0);
      for (int c = 0; c < 3; ++c) {
        prm_.x_ub[3 + c] = max_vel, prm_.x_lb[3 + c] = -max_vel;
        prm_.x_ub[6 + c] = c < 2 ? max_acc_xy : max_acc_z, prm_.x_lb[6 + c] = c < 2 ? min_acc_xy : min_acc_z;
        prm_.u_ub[c] = max_jerk, prm_.u_lb[c] = -max_jerk;
      }
    }
    declare_parameter("mass", 1.0);
    yaw_idx_ = declare_parameter("yaw_idx", 3);
    k_p_yaw_ = declare_parameter("k_p_yaw", 1.0);
    const std::vector<double> drag = declare_parameter("drag_coeff", std::vector<double>(3, 0.0));
    for (int c = 0; c < 3; ++c) prm_.drag[c] = c < (int)drag.size() ? drag[c] : 0.0;
    const std::vector<double> ini = declare_parameter("state_ini", std::vector<double>(3 * n_local_, 0.0));  // [n_local][3]
    const std::vector<double> goal = declare_parameter("goal", std::vector<double>(3 * n_local_, 0.0));
    declare_parameter("planner_verbose", false);
    save_stats_ = declare_parameter("save_stats", false);
    declare_parameter("dmp_search_rad", 0.0);
    declare_parameter("dmp_n_it", 1);
    declare_parameter("path_planning_period", 0.1);
    declare_parameter("remove_corners", false);
    for (int c = 0; c < 3; ++c) cfg_.grid_range[c] = c < (int)vg_range.size() ? vg_range[c] : cfg_.grid_range[c];
    if (yaw_idx_ > n_hor_) yaw_idx_ = n_hor_;
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
    yaw_.assign(n_local_, 0.0);
    hist_.resize(n_local_);
    for (int k = 0; k < n_local_; ++k) {  // AC:42-86
      const std::string base = topic_ + "_" + std::to_string(first_ + k);
      pubs_.push_back(create_publisher<Trajectory>(base + "/traj_full", 10));
      traj_pubs_.push_back(create_publisher<Path>(base + "/traj", 10));
      traj_ref_pubs_.push_back(create_publisher<Path>(base + "/traj_ref", 10));
      path_pubs_.push_back(create_publisher<Path>(base + "/path", 10));
      hist_pubs_.push_back(create_publisher<Path>(base + "/traj_hist", 10));
      poly_pubs_.push_back(create_publisher<decomp_ros_msgs::msg::PolyhedronArray>(base + "/polyhedra", 10));
      seed_pubs_.push_back(create_publisher<sensor_msgs::msg::PointCloud2>(base + "/seeds", 10));
      pos_pubs_.push_back(create_publisher<visualization_msgs::msg::Marker>(base + "/position", 10));
    }
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
  double yaw(int k) const { return yaw_[k]; }

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
    hdsm_swarm_yaw(swarm_, yaw_idx_, k_p_yaw_, yaw_.data());  // ComputeYawAngle, AC:177 (after the solve, before the state advances)
    hdsm_swarm_commit(swarm_, traj_.data(), ctrl_.data(), used_.data(), status_.data(), local_.data(), has_local_.data());
    std::lock_guard<std::mutex> g(mtx_);
    for (int k = 0; k < n_local_; ++k) {
      if (!has_local_[k]) continue;
      std::copy_n(&local_[(size_t)k * (N + 1) * 9], (N + 1) * 9, &plans_[(size_t)(first_ + k) * (N + 1) * 9]);
      has_[first_ + k] = 1;
      Trajectory msg;  // PublishTrajectoryFull, AC:645-677
      msg.stamp = now();
      msg.yaw = yaw_[k];
      msg.dt = prm_.dt;
      for (int i = 0; i <= N; ++i) {
        const double* rec = &local_[((size_t)k * (N + 1) + i) * 9];
        multi_agent_planner_msgs::msg::State st;
        st.position = {rec[0], rec[1], rec[2]}, st.velocity = {rec[3], rec[4], rec[5]}, st.acceleration = {rec[6], rec[7], rec[8]};
        msg.states.push_back(st);
      }
      pubs_[k]->publish(msg);
    }
    for (int k = 0; k < n_local_; ++k) publish_rviz(k);  // AC:246-258
    ++rounds_;
  }

  Path path_msg(const double* pts, int n) {
    Path m;
    m.header.stamp = now(), m.header.frame_id = world_frame_;
    for (int i = 0; i < n; ++i) {
      geometry_msgs::msg::PoseStamped ps;
      ps.pose.position.x = pts[3 * i], ps.pose.position.y = pts[3 * i + 1], ps.pose.position.z = pts[3 * i + 2];
      m.poses.push_back(ps);
    }
    return m;
  }

  void publish_rviz(int k) {  // PublishCurrentPosition / Trajectory / ReferencePath / Path / PolyhedraSeeds / Polyhedra / TrajectoryHistory
    const int N = n_hor_, P = prm_.poly_hor, RS = prm_.max_rows_static;
    constexpr int PMAX = 64;
    std::vector<double> tc(3 * (N + 1)), tr(3 * (N + 1)), pth(3 * PMAX), pA((size_t)P * RS * 3), pb((size_t)P * RS), seeds(3 * P);
    std::vector<int32_t> rows(P, 0);
    int32_t n_traj = 0, n_ref = 0, n_path = 0, n_poly = 0;
    double pos[3];
    if (hdsm_swarm_view(swarm_, k, tc.data(), &n_traj, tr.data(), &n_ref, pth.data(), PMAX, &n_path, &n_poly, rows.data(), pA.data(), pb.data(),
                        seeds.data(), pos) != HDSM_OK)
      return;
    {  // AC:806-826
      visualization_msgs::msg::Marker mk;
      mk.header.frame_id = world_frame_, mk.header.stamp = now();
      mk.type = visualization_msgs::msg::Marker::SPHERE, mk.action = visualization_msgs::msg::Marker::ADD;
      mk.pose.position.x = pos[0], mk.pose.position.y = pos[1], mk.pose.position.z = pos[2], mk.pose.orientation.w = 1.0;
      mk.scale.x = 2 * prm_.drone_radius, mk.scale.y = 2 * prm_.drone_radius, mk.scale.z = 2 * prm_.drone_z_offset;
      mk.color.a = 1, mk.color.r = 0.0, mk.color.g = 1.0, mk.color.b = 0.0;
      pos_pubs_[k]->publish(mk);
    }
    if (n_traj > 0) traj_pubs_[k]->publish(path_msg(tc.data(), n_traj));  // AC:679-700
    traj_ref_pubs_[k]->publish(path_msg(tr.data(), n_ref));               // AC:757-780
    path_pubs_[k]->publish(path_msg(pth.data(), n_path));                  // AC:702-729
    {  // AC:829-849: the seeds as an unorganised xyz cloud (what pcl::toROSMsg writes for PointXYZ: 16-byte points, float32 x y z)
      sensor_msgs::msg::PointCloud2 pc;
      pc.header.frame_id = world_frame_;
      pc.height = 1, pc.width = (uint32_t)n_poly, pc.point_step = 16, pc.row_step = 16 * (uint32_t)n_poly, pc.is_dense = true;
      const char* nm[3] = {"x", "y", "z"};
      for (int c = 0; c < 3; ++c) {
        sensor_msgs::msg::PointField f;
        f.name = nm[c], f.offset = 4 * c, f.datatype = sensor_msgs::msg::PointField::FLOAT32, f.count = 1;
        pc.fields.push_back(f);
      }
      pc.data.assign((size_t)16 * n_poly, 0);
      for (int j = 0; j < n_poly; ++j)
        for (int c = 0; c < 3; ++c) {
          const float v = (float)seeds[3 * j + c];
          std::memcpy(&pc.data[(size_t)16 * j + 4 * c], &v, 4);
        }
      seed_pubs_[k]->publish(pc);
    }
    {  // AC:851-857: DecompROS::polyhedron_array_to_ros — one (point, outer normal) pair per hyperplane; a point of row n . x <= b is n b / |n|^2
      decomp_ros_msgs::msg::PolyhedronArray pa;
      pa.header.frame_id = world_frame_;
      for (int j = 0; j < n_poly; ++j) {
        decomp_ros_msgs::msg::Polyhedron ph;
        for (int r = 0; r < rows[j]; ++r) {
          const double* n = &pA[((size_t)j * RS + r) * 3];
          const double nn = n[0] * n[0] + n[1] * n[1] + n[2] * n[2], bb = pb[(size_t)j * RS + r];
          geometry_msgs::msg::Point pt, nv;
          if (nn > 0) pt.x = n[0] * bb / nn, pt.y = n[1] * bb / nn, pt.z = n[2] * bb / nn;
          nv.x = n[0], nv.y = n[1], nv.z = n[2];
          ph.points.push_back(pt), ph.normals.push_back(nv);
        }
        pa.polyhedrons.push_back(ph);
      }
      poly_pubs_[k]->publish(pa);
    }
    hist_[k].insert(hist_[k].end(), pos, pos + 3);  // state_hist_ (AC:240-245); the published history leaves out the newest entry (AC:739)
    const int nh = (int)hist_[k].size() / 3;
    hist_pubs_[k]->publish(path_msg(hist_[k].data(), nh > 0 ? nh - 1 : 0));
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
  int yaw_idx_ = 3;
  double k_p_yaw_ = 1.0;
  std::vector<double> yaw_;
  std::vector<std::vector<double>> hist_;
  std::string topic_, world_frame_;
  std::mutex mtx_;
  std::vector<double> plans_, state_, ref_, A_, b_, traj_, ctrl_, obj_, local_;
  std::vector<uint8_t> has_, used_, has_local_;
  std::vector<int32_t> id_, n_poly_, n_rows_, status_;
  std::vector<void*> stats_remote_;
  std::vector<Lat> latency_;
  std::vector<rclcpp::Publisher<Trajectory>::SharedPtr> pubs_;
  std::vector<rclcpp::Publisher<Path>::SharedPtr> traj_pubs_, traj_ref_pubs_, path_pubs_, hist_pubs_;
  std::vector<rclcpp::Publisher<decomp_ros_msgs::msg::PolyhedronArray>::SharedPtr> poly_pubs_;
  std::vector<rclcpp::Publisher<sensor_msgs::msg::PointCloud2>::SharedPtr> seed_pubs_;
  std::vector<rclcpp::Publisher<visualization_msgs::msg::Marker>::SharedPtr> pos_pubs_;
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
