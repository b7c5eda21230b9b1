// UVO_node_hip.cpp -- the ROS-1 adapter: UVO's node with the MI355X hot path behind it.
//
// Built ONLY where roscpp exists (catkin; see CMakeLists.txt next to this file) -- this repository's image has no ROS, so the
// file is never compiled here; everything it calls (the parameter loaders, the node class, the uvo_libraries surface) is
// compiled and tested without ROS (tests/test_node.py, tests/test_shim.py).  It keeps the reference's external surface:
//   node name            UVO_node                                              (uvo/src/UVO_node.cpp:11)
//   parameters           /visual_odometry_node ("mono" | "stereo"), /camera_name, and the keys of uvo/config/*.yaml
//                                                                              (UVO_node.cpp:23, visual_odometry.h:756, VO_utility.cpp:387-507)
//   subscriptions        stereo: /image_left/compressed, /image_right/compressed (sensor_msgs/CompressedImage, queue 1,
//                        ApproximateTime policy with queue 10); mono: /image/compressed (queue 1), /range (sensor_msgs/Range, queue 1)
//                                                                              (visual_odometry.h:766-774, 784-785)
//   publications         /estimated_linear_vel_{stereo,mono}_UVO (geometry_msgs/Vector3Stamped), /validity_{stereo,mono}_UVO
//                        (std_msgs/Bool), queue 10                             (visual_odometry.h:763-764, 781-782)
//   loop                 ros::Rate(NODE_FREQ): spinOnce, sleep, one loop body   (visual_odometry.h:249-251, 528-530)
// Image decoding stays where the reference has it (cv_bridge::toCvCopy + COLOR_BayerBGGR2BGR for bayer formats,
// uvo_libraries/src/math_utility.cpp:154-173) when OpenCV is present; without it the library's own decoder is used
// (uvo_hip::decode_compressed_image, include/uvo_libraries_hip/image_codec.h).
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
#include <memory>

#include "uvo_libraries_hip/visual_odometry_hip.h"
#include "uvo_libraries_hip/image_codec.h"
#ifdef UVO_HAVE_OPENCV
#include <cv_bridge/cv_bridge.h>
#include <opencv2/imgproc.hpp>
#endif

namespace {

// the parameter server's subtree -> ParamTree (same conversions afterwards as with the YAML loader)
void copy_param(const std::string& key, const XmlRpc::XmlRpcValue& v, uvo_hip::ParamTree& tree)
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

uvocv::Mat decode(const sensor_msgs::CompressedImage::ConstPtr& msg)
{
#ifdef UVO_HAVE_OPENCV
    cv_bridge::CvImagePtr cv_ptr = cv_bridge::toCvCopy(msg);                                   // MU:160
    if (msg->format.find("bayer") != std::string::npos) cv::cvtColor(cv_ptr->image, cv_ptr->image, cv::COLOR_BayerBGGR2BGR);   // MU:161-164
    return cv_ptr->image;
#else
    return uvo_hip::decode_compressed_image(msg->data.data(), msg->data.size(), msg->format);
#endif
}

struct Adapter {
    ros::NodeHandle nh;
    std::unique_ptr<uvo_hip::visual_odometry_core> core;
    ros::Publisher pub_vel, pub_valid;
    void mono_cb(const sensor_msgs::CompressedImage::ConstPtr& msg) { core->mono_imgs_callback(decode(msg), msg->header.stamp.toSec()); }
    void range_cb(const sensor_msgs::Range::ConstPtr& msg) { core->range_callback(msg->range); }
    void stereo_cb(const sensor_msgs::CompressedImage::ConstPtr& l, const sensor_msgs::CompressedImage::ConstPtr& r)
    { core->stereo_imgs_callback(decode(l), decode(r), l->header.stamp.toSec()); }
    void publish(const uvo_hip::Published& p)
    {
        if (!p.published) return;
        geometry_msgs::Vector3Stamped v;
        v.header.stamp.sec = ros::Time::now().toSec();                                           // visual_odometry.h:129 / 150
        v.vector.x = p.v[0]; v.vector.y = p.v[1]; v.vector.z = p.v[2];
        std_msgs::Bool ok; ok.data = p.valid;
        pub_vel.publish(v); pub_valid.publish(ok);
    }
};

}  // namespace

int main(int argc, char** argv)
{
    ros::init(argc, argv, "UVO_node");
    Adapter a;
    std::string VO_NODE, CAMERA_NAME;
    ros::Rate wait(20);
    while (ros::ok() && !ros::param::get("/visual_odometry_node", VO_NODE)) { ros::spinOnce(); wait.sleep(); }       // UVO_node.cpp:18-23
    a.nh.getParam("/camera_name", CAMERA_NAME);                                                                       // visual_odometry.h:756
    uvo_hip::ParamTree tree;
    XmlRpc::XmlRpcValue root;
    if (a.nh.getParam("/", root)) copy_param("", root, tree);
    try {
        a.core.reset(new uvo_hip::visual_odometry_core(VO_NODE, tree, CAMERA_NAME));
    } catch (const std::exception& e) { ROS_ERROR("%s", e.what()); return 1; }
    ros::Rate loop_rate(NODE_FREQ);                                                                                   // visual_odometry.h:759
    typedef message_filters::sync_policies::ApproximateTime<sensor_msgs::CompressedImage, sensor_msgs::CompressedImage> SyncPolicy;
    std::unique_ptr<message_filters::Subscriber<sensor_msgs::CompressedImage>> sub_l, sub_r;
    std::unique_ptr<message_filters::Synchronizer<SyncPolicy>> sync;
    ros::Subscriber sub_img, sub_range;
    if (VO_NODE == "stereo") {
        a.pub_vel = a.nh.advertise<geometry_msgs::Vector3Stamped>("/estimated_linear_vel_stereo_UVO", 10);
        a.pub_valid = a.nh.advertise<std_msgs::Bool>("/validity_stereo_UVO", 10);
        sub_l.reset(new message_filters::Subscriber<sensor_msgs::CompressedImage>(a.nh, "/image_left/compressed", 1));
        sub_r.reset(new message_filters::Subscriber<sensor_msgs::CompressedImage>(a.nh, "/image_right/compressed", 1));
        sync.reset(new message_filters::Synchronizer<SyncPolicy>(SyncPolicy(10), *sub_l, *sub_r));
        sync->registerCallback(boost::bind(&Adapter::stereo_cb, &a, _1, _2));
    } else {
        a.pub_vel = a.nh.advertise<geometry_msgs::Vector3Stamped>("/estimated_linear_vel_mono_UVO", 10);
        a.pub_valid = a.nh.advertise<std_msgs::Bool>("/validity_mono_UVO", 10);
        sub_img = a.nh.subscribe("/image/compressed", 1, &Adapter::mono_cb, &a);
        sub_range = a.nh.subscribe("/range", 1, &Adapter::range_cb, &a);
    }
    while (ros::ok()) {
        ros::spinOnce();
        loop_rate.sleep();
        try { a.publish(a.core->spin_once()); }
        catch (const std::exception& e) { ROS_ERROR("UVO hot path: %s", e.what()); return 1; }      // the reference dies on cv::Exception and is respawned (launch:24,38)
    }
    return 0;
}
