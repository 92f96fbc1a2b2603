// oracle/cvstub: the reference translation units on the hot path include <ros/ros.h> but use nothing from it.
