// NOT ROS (see README.md)
#pragma once
#include <ros/ros.h>
namespace geometry_msgs { struct Vector3 { double x = 0, y = 0, z = 0; }; struct Vector3Stamped { std_msgs::Header header; Vector3 vector; }; }
