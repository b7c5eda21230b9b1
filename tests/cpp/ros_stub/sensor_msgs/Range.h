// NOT ROS (see README.md)
#pragma once
#include <memory>
#include <ros/ros.h>
namespace sensor_msgs { struct Range { std_msgs::Header header; float range = 0; typedef std::shared_ptr<Range const> ConstPtr; }; }
