#!/bin/bash
# round 4, session j: the ROS-signature adapter behind std::unique_ptr<Filter> on golden streams; the whole GPU suite
timeout 600 python -m pytest tests/test_ros_adapter.py -q -x 2>&1 | tail -15
timeout 2400 python -m pytest tests -q -m gpu 2>&1 | tail -8
