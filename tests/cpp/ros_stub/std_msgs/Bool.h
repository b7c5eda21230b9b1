// NOT ROS (see README.md)
#pragma once
namespace std_msgs { struct Bool { bool data = false; }; }
