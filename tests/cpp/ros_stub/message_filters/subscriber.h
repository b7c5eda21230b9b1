// NOT ROS (see README.md)
#pragma once
#include <ros/ros.h>
namespace message_filters { template <class M> struct Subscriber { Subscriber(ros::NodeHandle& nh, const std::string& topic, uint32_t queue_size); }; }
