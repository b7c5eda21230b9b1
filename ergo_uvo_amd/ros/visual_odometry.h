// visual_odometry.h -- `class visual_odometry_node` of the reference (uvo/include/visual_odometry.h:35-113) with the MI355X hot path
// behind it, under the reference's header name, so that the reference's own main() (uvo/src/UVO_node.cpp:9-29:
// `#include <visual_odometry.h>`, `visual_odometry_node visual_odometry_node_object;`, `...visual_odometry_workflow(VO_NODE);`)
// compiles against it UNCHANGED -- put this directory in front of the reference's uvo/include on the include path and link
// uvo_libraries_hip + uvo_hip instead of uvo_libraries.
//
// Built ONLY where roscpp exists -- this repository's image has no ROS, so the header is never compiled into a node here; it is
// parsed and type-checked against tests/cpp/ros_stub (tests/test_node.py), and everything it calls (parameter loaders, the
// loops, the uvo_libraries surface) is compiled and tested without ROS (tests/test_node.py, tests/test_shim.py).
// The external surface is the reference's:
//   parameters      /camera_name and the keys of uvo/config/*.yaml                        (visual_odometry.h:756-757, VO_utility.cpp:387-507)
//   subscriptions   stereo: /image_left/compressed, /image_right/compressed (sensor_msgs/CompressedImage, queue 1, ApproximateTime
//                   policy with queue 10); mono: /image/compressed (queue 1), /range (sensor_msgs/Range, queue 1)
//                                                                                         (visual_odometry.h:766-774, 784-785)
//   publications    /estimated_linear_vel_{stereo,mono}_UVO (geometry_msgs/Vector3Stamped), /validity_{stereo,mono}_UVO
//                   (std_msgs/Bool), queue 10                                             (visual_odometry.h:763-764, 781-782)
//   loop            ros::Rate(NODE_FREQ): spinOnce, sleep, one loop body                  (visual_odometry.h:249-251, 528-530)
// Image decoding stays where the reference has it (cv_bridge::toCvCopy + COLOR_BayerBGGR2BGR for bayer formats,
// uvo_libraries/src/math_utility.cpp:154-173) when OpenCV is present; without it the library's own decoder is used
// (uvo_hip::decode_compressed_image, include/uvo_libraries_hip/image_codec.h).
#pragma once
#include <ros/ros.h>
#include <XmlRpcValue.h>
#include <geometry_msgs/Vector3Stamped.h>
#include <message_filters/subscriber.h>
#include <message_filters/sync_policies/approximate_time.h>
#include <message_filters/synchronizer.h>
#include <sensor_msgs/CompressedImage.h>
#include <sensor_msgs/Range.h>
#include <std_msgs/Bool.h>
#include <boost/bind.hpp>
#include <exception>
#include <memory>
#include <string>

#include "uvo_libraries_hip/visual_odometry_hip.h"
#include "uvo_libraries_hip/image_codec.h"
#ifdef UVO_HAVE_OPENCV
#include <cv_bridge/cv_bridge.h>
#include <opencv2/imgproc.hpp>
#endif

using namespace std;          // the reference's headers do (uvo_libraries/math_utility.h:13) and its main() relies on it: `string VO_NODE;`

class visual_odometry_node
{
    private:

        ros::NodeHandle node_obj;
        std::string CAMERA_NAME;
        std::unique_ptr<uvo_hip::visual_odometry_core> core;          // the two loops and what they keep between iterations
        ros::Publisher pub_estimated_linear_vel, pub_validity;

        // the parameter server's subtree -> ParamTree (same conversions afterwards as with the YAML loader)
        static void copy_param(const std::string& key, const XmlRpc::XmlRpcValue& v, uvo_hip::ParamTree& tree)
        {
            using V = uvo_hip::ParamTree::Value;
            V out;
            switch (v.getType()) {
            case XmlRpc::XmlRpcValue::TypeInt:     out.kind = V::kInt; out.i = (int)const_cast<XmlRpc::XmlRpcValue&>(v); out.d = (double)out.i; break;
            case XmlRpc::XmlRpcValue::TypeDouble:  out.kind = V::kDouble; out.d = (double)const_cast<XmlRpc::XmlRpcValue&>(v); break;
            case XmlRpc::XmlRpcValue::TypeBoolean: out.kind = V::kBool; out.b = (bool)const_cast<XmlRpc::XmlRpcValue&>(v); break;
            case XmlRpc::XmlRpcValue::TypeString:  out.kind = V::kString; out.s = (std::string)const_cast<XmlRpc::XmlRpcValue&>(v); break;
            case XmlRpc::XmlRpcValue::TypeArray:
                out.kind = V::kList;
                for (int i = 0; i < v.size(); i++) {
                    XmlRpc::XmlRpcValue e = v[i];
                    if (e.getType() == XmlRpc::XmlRpcValue::TypeInt) out.list.push_back((double)(int)e);
                    else if (e.getType() == XmlRpc::XmlRpcValue::TypeDouble) out.list.push_back((double)e);
                }
                break;
            case XmlRpc::XmlRpcValue::TypeStruct: {
                XmlRpc::XmlRpcValue& s = const_cast<XmlRpc::XmlRpcValue&>(v);
                for (XmlRpc::XmlRpcValue::iterator it = s.begin(); it != s.end(); ++it) copy_param(key + "/" + it->first, it->second, tree);
                return;
            }
            default: return;
            }
            tree.set(key, out);
        }

        static uvocv::Mat from_ros_to_cv_image(const sensor_msgs::CompressedImage::ConstPtr& msg)       // math_utility.cpp:154-173
        {
#ifdef UVO_HAVE_OPENCV
            cv_bridge::CvImagePtr cv_ptr = cv_bridge::toCvCopy(msg);                                   // MU:160
            if (msg->format.find("bayer") != std::string::npos) cv::cvtColor(cv_ptr->image, cv_ptr->image, cv::COLOR_BayerBGGR2BGR);   // MU:161-164
            return cv_ptr->image;
#else
            return uvo_hip::decode_compressed_image(msg->data.data(), msg->data.size(), msg->format);
#endif
        }

        // the subscribers' callbacks (visual_odometry.h:67-78, 88-95): the newest message replaces an unprocessed one
        void mono_imgs_callback(const sensor_msgs::CompressedImage::ConstPtr& msg) { core->mono_imgs_callback(from_ros_to_cv_image(msg), msg->header.stamp.toSec()); }
        void range_callback(const sensor_msgs::Range::ConstPtr& msg) { core->range_callback(msg->range); }
        void stereo_imgs_callback(const sensor_msgs::CompressedImage::ConstPtr& left_image, const sensor_msgs::CompressedImage::ConstPtr& right_image)
        { core->stereo_imgs_callback(from_ros_to_cv_image(left_image), from_ros_to_cv_image(right_image), left_image->header.stamp.toSec()); }

        void publish(const uvo_hip::Published& p)                                                      // visual_odometry.h:121-133, 142-160
        {
            if (!p.published) return;
            geometry_msgs::Vector3Stamped v;
            v.header.stamp.sec = ros::Time::now().toSec();                                             // visual_odometry.h:129 / 150
            v.vector.x = p.v[0]; v.vector.y = p.v[1]; v.vector.z = p.v[2];
            std_msgs::Bool ok; ok.data = p.valid;
            pub_estimated_linear_vel.publish(v); pub_validity.publish(ok);
        }

        // mono_VO / stereo_VO (visual_odometry.h:167-398, 406-740): the loop bodies live in visual_odometry_core::spin_once()
        void VO_loop(ros::Rate loop_rate)
        {
            while (ros::ok()) {
                ros::spinOnce();
                loop_rate.sleep();
                publish(core->spin_once());       // a uvo_hip::Error leaves the node as a cv::Exception leaves the reference's: respawned by the launch file (launch:24,38)
            }
        }

    public:

        visual_odometry_node()                                                                         // visual_odometry.h:104-109
        {
            ROS_WARN(" ################## BUILDING THE OBJECT FOR THE VISUAL ODOMETRY TASK (MI355X hot path) ################## \n");
        }

        void visual_odometry_workflow(std::string VO_NODE);                                            // visual_odometry.h:112
};

// visual_odometry.h:749-791
inline void visual_odometry_node::visual_odometry_workflow(std::string VO_NODE)
{
    if (VO_NODE != "stereo" && VO_NODE != "mono") {                                                    // visual_odometry.h:789
        ROS_ERROR(" ################ WRONG SELECTION OF VISUAL ODOMETRY NODE - CHOOSE BETWEEN mono AND stereo ################");
        return;
    }
    node_obj.getParam("/camera_name", CAMERA_NAME);                                                    // visual_odometry.h:756
    uvo_hip::ParamTree tree;
    XmlRpc::XmlRpcValue root;
    if (node_obj.getParam("/", root)) copy_param("", root, tree);
    // get_VO_parameters + get_{stereo,mono}_camera_parameters (visual_odometry.h:757, 776, 787) run in the core's constructor
    core.reset(new uvo_hip::visual_odometry_core(VO_NODE, tree, CAMERA_NAME));
    ros::Rate loop_rate(NODE_FREQ);                                                                    // visual_odometry.h:759

    if (VO_NODE == "stereo") {
        pub_estimated_linear_vel = node_obj.advertise<geometry_msgs::Vector3Stamped>("/estimated_linear_vel_stereo_UVO", 10);
        pub_validity = node_obj.advertise<std_msgs::Bool>("/validity_stereo_UVO", 10);
        message_filters::Subscriber<sensor_msgs::CompressedImage> sub_cameraSX(node_obj, "/image_left/compressed", 1);
        message_filters::Subscriber<sensor_msgs::CompressedImage> sub_cameraDX(node_obj, "/image_right/compressed", 1);
        typedef message_filters::sync_policies::ApproximateTime<sensor_msgs::CompressedImage, sensor_msgs::CompressedImage> MySyncPolicy;
        message_filters::Synchronizer<MySyncPolicy> sync(MySyncPolicy(10), sub_cameraSX, sub_cameraDX);
        sync.registerCallback(boost::bind(&visual_odometry_node::stereo_imgs_callback, this, _1, _2));
        VO_loop(loop_rate);
    } else {
        pub_estimated_linear_vel = node_obj.advertise<geometry_msgs::Vector3Stamped>("/estimated_linear_vel_mono_UVO", 10);
        pub_validity = node_obj.advertise<std_msgs::Bool>("/validity_mono_UVO", 10);
        ros::Subscriber sub_camera_imgs = node_obj.subscribe("/image/compressed", 1, &visual_odometry_node::mono_imgs_callback, this);
        ros::Subscriber sub_range = node_obj.subscribe("/range", 1, &visual_odometry_node::range_callback, this);
        VO_loop(loop_rate);
    }
}
