// NOT ROS: declarations only, for a syntax check of ergo_uvo_amd/ros/UVO_node_hip.cpp (tests/cpp/ros_stub/README.md)
#pragma once
#include <cstdint>
#include <cstdio>
#include <memory>
#include <string>
#include "XmlRpcValue.h"
#define ROS_ERROR(...) std::fprintf(stderr, __VA_ARGS__)
#define ROS_WARN(...) std::fprintf(stderr, __VA_ARGS__)
namespace ros {
void init(int& argc, char** argv, const std::string& name);
bool ok();
void spinOnce();
struct Time { uint32_t sec = 0, nsec = 0; double toSec() const; static Time now(); };
struct Rate { explicit Rate(double hz); bool sleep(); };
struct Publisher { template <class M> void publish(const M& m) const; };
struct Subscriber {};
struct NodeHandle {
    template <class T> bool getParam(const std::string& key, T& value) const;
    template <class M> Publisher advertise(const std::string& topic, uint32_t queue_size);
    template <class M, class T> Subscriber subscribe(const std::string& topic, uint32_t queue_size, void (T::*fp)(const std::shared_ptr<M const>&), T* obj);
};
namespace param { template <class T> bool get(const std::string& key, T& value); }
}  // namespace ros
namespace std_msgs { struct Header { uint32_t seq = 0; ros::Time stamp; std::string frame_id; }; }
