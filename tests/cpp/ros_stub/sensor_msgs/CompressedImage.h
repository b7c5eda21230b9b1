// NOT ROS (see README.md)
#pragma once
#include <memory>
#include <string>
#include <vector>
#include <ros/ros.h>
namespace sensor_msgs { struct CompressedImage { std_msgs::Header header; std::string format; std::vector<uint8_t> data; typedef std::shared_ptr<CompressedImage const> ConstPtr; }; }
