#pragma once
#include <memory>
namespace rclcpp_lifecycle { struct LifecycleNode { typedef std::shared_ptr<LifecycleNode> SharedPtr; }; }
