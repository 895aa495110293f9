// TEST INFRASTRUCTURE — sensor_msgs/msg/PointCloud2 and PointField: the fields of the ROS 2 definitions.
#pragma once
#include <cstdint>
#include <memory>
#include <std_msgs/msg/header.hpp>
#include <vector>
namespace sensor_msgs {
namespace msg {
struct PointField {
  static constexpr uint8_t FLOAT32 = 7;
  std::string name;
  uint32_t offset = 0;
  uint8_t datatype = 0;
  uint32_t count = 0;
};
struct PointCloud2 {
  std_msgs::msg::Header header;
  uint32_t height = 0, width = 0;
  std::vector<PointField> fields;
  bool is_bigendian = false;
  uint32_t point_step = 0, row_step = 0;
  std::vector<uint8_t> data;
  bool is_dense = false;
  using SharedPtr = std::shared_ptr<PointCloud2>;
};
}  // namespace msg
}  // namespace sensor_msgs
