// TEST INFRASTRUCTURE — an API-shaped stand-in for <rclcpp/rclcpp.hpp>, just large enough to TYPE-CHECK ros/hdsm_agent_node.cpp
// and to run nodes in one process: parameters with per-node overrides, publishers / subscriptions on an in-memory bus
// (synchronous delivery to every subscription of the topic, like an intra-process DDS), wall timers fired by spin_some(),
// a clock, on_shutdown hooks. ROS 2 is not installed in the build image; nothing here is product code and the product never
// includes it. Signatures follow rclcpp (Humble): Node::declare_parameter<T>(name, default), create_publisher<T>(topic, qos),
// create_subscription<T>(topic, qos, callback(const T::SharedPtr)), create_wall_timer(period, callback), now(), get_logger().
#pragma once
#include <builtin_interfaces/msg/time.hpp>

#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <functional>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <utility>
#include <variant>
#include <vector>

namespace rclcpp {

class Time {
 public:
  Time() = default;
  explicit Time(int64_t ns) : ns_(ns) {}
  Time(const builtin_interfaces::msg::Time& t) : ns_((int64_t)t.sec * 1000000000LL + t.nanosec) {}  // NOLINT (rclcpp converts implicitly)
  operator builtin_interfaces::msg::Time() const {                                                  // NOLINT
    builtin_interfaces::msg::Time t;
    t.sec = (int32_t)(ns_ / 1000000000LL), t.nanosec = (uint32_t)(ns_ % 1000000000LL);
    return t;
  }
  int64_t nanoseconds() const { return ns_; }
  double seconds() const { return (double)ns_ * 1e-9; }

 private:
  int64_t ns_ = 0;
};
class Duration {
 public:
  explicit Duration(int64_t ns) : ns_(ns) {}
  double seconds() const { return (double)ns_ * 1e-9; }

 private:
  int64_t ns_;
};
inline Duration operator-(const Time& a, const Time& b) { return Duration(a.nanoseconds() - b.nanoseconds()); }

class Logger {};
#define RCLCPP_INFO(logger, ...)          \
  do {                                    \
    (void)(logger);                       \
    std::fprintf(stderr, "[INFO] ");      \
    std::fprintf(stderr, __VA_ARGS__);    \
    std::fprintf(stderr, "\n");           \
  } while (0)

using ParameterValue = std::variant<bool, int, double, std::string, std::vector<double>>;

namespace shim {
struct Bus {
  // topic -> callbacks taking a type-erased shared_ptr to the message
  std::map<std::string, std::vector<std::function<void(std::shared_ptr<void>)>>> subs;
  std::map<std::string, long> published, delivered;
  std::vector<std::function<void()>> timers, shutdown_hooks;
  std::map<std::string, ParameterValue> overrides;  // applied to the NEXT node that is constructed
  int64_t clock_ns = 1000000000LL;
  bool ok = false;
};
inline Bus& bus() {
  static Bus b;
  return b;
}
}  // namespace shim

template <class MessageT>
class Publisher {
 public:
  using SharedPtr = std::shared_ptr<Publisher<MessageT>>;
  explicit Publisher(std::string topic) : topic_(std::move(topic)) {}
  void publish(const MessageT& msg) {
    shim::Bus& b = shim::bus();
    ++b.published[topic_];
    auto it = b.subs.find(topic_);
    if (it == b.subs.end()) return;
    auto copy = std::make_shared<MessageT>(msg);
    for (auto& cb : it->second) {
      cb(copy);
      ++b.delivered[topic_];
    }
  }

 private:
  std::string topic_;
};
template <class MessageT>
class Subscription {
 public:
  using SharedPtr = std::shared_ptr<Subscription<MessageT>>;
};
class TimerBase {
 public:
  using SharedPtr = std::shared_ptr<TimerBase>;
};

class Node {
 public:
  explicit Node(const std::string& name) : name_(name), overrides_(std::move(shim::bus().overrides)) { shim::bus().overrides.clear(); }
  virtual ~Node() = default;
  template <class T>
  T declare_parameter(const std::string& name, const T& def) {
    auto it = overrides_.find(name);
    if (it == overrides_.end()) return def;
    if (const T* v = std::get_if<T>(&it->second)) return *v;
    throw std::runtime_error("parameter '" + name + "' overridden with another type");
  }
  std::string declare_parameter(const std::string& name, const char* def) { return declare_parameter<std::string>(name, std::string(def)); }
  template <class MessageT>
  typename Publisher<MessageT>::SharedPtr create_publisher(const std::string& topic, int /*qos depth*/) {
    return std::make_shared<Publisher<MessageT>>(topic);
  }
  template <class MessageT, class CallbackT>
  typename Subscription<MessageT>::SharedPtr create_subscription(const std::string& topic, int /*qos depth*/, CallbackT&& cb) {
    std::function<void(const typename MessageT::SharedPtr)> f = std::forward<CallbackT>(cb);
    shim::bus().subs[topic].push_back([f](std::shared_ptr<void> m) { f(std::static_pointer_cast<MessageT>(m)); });
    return std::make_shared<Subscription<MessageT>>();
  }
  template <class Rep, class Period, class CallbackT>
  TimerBase::SharedPtr create_wall_timer(std::chrono::duration<Rep, Period> /*period*/, CallbackT&& cb) {
    shim::bus().timers.push_back(std::function<void()>(std::forward<CallbackT>(cb)));
    return std::make_shared<TimerBase>();
  }
  Time now() const { return Time(shim::bus().clock_ns); }
  Logger get_logger() const { return Logger(); }
  const std::string& get_name() const { return name_; }

 private:
  std::string name_;
  std::map<std::string, ParameterValue> overrides_;
};

inline void init(int, char**) { shim::bus().ok = true; }
inline bool ok() { return shim::bus().ok; }
template <class F>
void on_shutdown(F&& f) {
  shim::bus().shutdown_hooks.push_back(std::function<void()>(std::forward<F>(f)));
}
// one executor pass: every timer fires once, the clock advances by `dt_ns`
inline void spin_some_all(int64_t dt_ns = 100000000LL) {
  shim::Bus& b = shim::bus();
  for (size_t k = 0; k < b.timers.size(); ++k) b.timers[k]();
  b.clock_ns += dt_ns;
}
inline void spin(std::shared_ptr<Node> /*node*/) {  // the real spin blocks until shutdown; here: a bounded number of passes
  const char* e = std::getenv("RCLCPP_SHIM_TICKS");
  const int ticks = e ? std::atoi(e) : 10;
  for (int k = 0; k < ticks && ok(); ++k) spin_some_all();
}
inline void shutdown() {
  shim::Bus& b = shim::bus();
  if (!b.ok) return;
  b.ok = false;
  for (auto& f : b.shutdown_hooks) f();
  b.shutdown_hooks.clear();
}

}  // namespace rclcpp
