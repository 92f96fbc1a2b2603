// oracle/matcherstub: ORBmatcher.cc includes <ros/ros.h> for ROS_ASSERT only.
#ifndef ORB_ORACLE_MATCHERSTUB_ROS_H
#define ORB_ORACLE_MATCHERSTUB_ROS_H
#include <stdexcept>
#define ROS_ASSERT(cond) do { if (!(cond)) throw std::runtime_error("ROS_ASSERT failed: " #cond); } while (0)
#endif
