// NOT ROS (see README.md)
#pragma once
namespace message_filters { namespace sync_policies { template <class M0, class M1> struct ApproximateTime { explicit ApproximateTime(unsigned queue_size); }; } }
