// NOT ROS (see README.md)
#pragma once
namespace message_filters { template <class Policy> struct Synchronizer { template <class F0, class F1> Synchronizer(const Policy& p, F0& f0, F1& f1); template <class C> void registerCallback(const C& callback); }; }
