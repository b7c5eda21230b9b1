// uvo_config.h -- the reference's parameter surface without ROS: the keys get_VO_parameters / get_mono_camera_parameters /
// get_stereo_camera_parameters read from the ROS parameter server (uvo_libraries/src/VO_utility.cpp:387-507), loaded from
// the same YAML files (uvo/config/*.yaml) into the same globals (uvo_libraries/include/uvo_libraries/VO_utility.h:21-89).
//
// ParamTree is the subset of the parameter server the node uses: a tree of scalars / lists addressed by "/a/b/c" keys, with
// roscpp's getParam() conversions (ros::param::getImpl): an int parameter accepts a double value rounded half away from
// zero at .5 (`distance: 10.0` -> DISTANCE = 10), a double accepts an int, bool and string need their own type, a missing
// key or a type mismatch leaves the variable untouched and returns false.  With ROS present the same tree is filled from a
// ros::NodeHandle (ergo_uvo_amd/ros/UVO_node_hip.cpp), so both paths end in the same globals.
#pragma once
#include <map>
#include <string>
#include <vector>
#include "uvo_libraries_hip/VO_utility_hip.h"

// ---- the remaining globals of VO_utility.h:21-39, 79-80 (the hot-path ones are declared in VO_utility_hip.h) ----
extern uvocv::Mat R_left, t_left, R_right, t_right;                                      // VOH:21
extern double fx, fy, ccx, ccy, k1, k2, p1, p2;                                          // VOH:27-28 (mono camera)
extern double fx_left, fy_left, ccx_left, ccy_left, fx_right, fy_right, ccx_right, ccy_right;      // VOH:31
extern double k1_left, k2_left, p1_left, p2_left, k1_right, k2_right, p1_right, p2_right;          // VOH:32
extern int    NODE_FREQ;                                                                 // VOH:39
extern int    FPS;  extern bool SHOW_MATCHES;                                            // VOH:79-80

namespace uvo_hip {

class ParamTree {
public:
    struct Value { enum Kind { kInt, kDouble, kBool, kString, kList } kind = kString; long long i = 0; double d = 0; bool b = false; std::string s; std::vector<double> list; };
    // YAML subset of the shipped files: nested mappings by indentation, `key: scalar`, `key: [a, b, c]`, `#` comments,
    // single- or double-quoted strings.  Keys are stored as "/outer/inner" under `ns` (roslaunch's <rosparam> namespace).
    void load_yaml_text(const std::string& text, const std::string& ns = "");
    void load_yaml_file(const std::string& path, const std::string& ns = "");            // throws uvo_hip::Error when unreadable
    void set(const std::string& key, const Value& v) { values_[key] = v; }
    bool has(const std::string& key) const { return values_.count(key) != 0; }
    bool getParam(const std::string& key, int& out) const;
    bool getParam(const std::string& key, double& out) const;
    bool getParam(const std::string& key, bool& out) const;
    bool getParam(const std::string& key, std::string& out) const;
    bool getParam(const std::string& key, std::vector<double>& out) const;
    size_t size() const { return values_.size(); }
private:
    std::map<std::string, Value> values_;
};

}  // namespace uvo_hip

// Same names and effects as the reference's loaders, reading a ParamTree instead of a ros::NodeHandle.
void get_VO_parameters(const uvo_hip::ParamTree& node_obj);                                          // VOU:461-507
void get_mono_camera_parameters(const uvo_hip::ParamTree& node_obj, std::string CAMERA_NAME);        // VOU:385-398
void get_stereo_camera_parameters(const uvo_hip::ParamTree& node_obj, std::string CAMERA_NAME);      // VOU:406-453
