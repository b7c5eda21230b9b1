// UVO_node_hip.cpp -- the node's bootstrap (uvo/src/UVO_node.cpp:9-29) over this directory's visual_odometry.h.
//
// The reference's own UVO_node.cpp compiles against that header unchanged (same class, constructor and workflow method:
// tests/test_node.py compiles it where it lies); this file is the same bootstrap for a workspace that does not carry the
// reference's sources, with one difference: it waits for /visual_odometry_node to exist before it enters the workflow, and
// turns an error of the hot path into an exit code for the launch file's respawn instead of an uncaught exception.
// Built only where roscpp exists (CMakeLists.txt next to this file).
#include "visual_odometry.h"

int main(int argc, char** argv)
{
    ros::init(argc, argv, "UVO_node");                                                     // UVO_node.cpp:11
    ros::NodeHandle main_node_obj;
    ros::Rate loop_rate(20);                                                               // UVO_node.cpp:14
    visual_odometry_node visual_odometry_node_object;                                      // UVO_node.cpp:16
    std::string which;
    while (ros::ok()) {
        ros::spinOnce();
        loop_rate.sleep();
        if (!ros::param::get("/visual_odometry_node", which)) continue;                    // UVO_node.cpp:23 (the reference enters with an empty string)
        try { visual_odometry_node_object.visual_odometry_workflow(which); }                // UVO_node.cpp:25
        catch (const std::exception& e) { ROS_ERROR("UVO hot path: %s", e.what()); return 1; }
    }
    return 0;
}
