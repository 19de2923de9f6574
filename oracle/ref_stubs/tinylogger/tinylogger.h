// Stand-in for the reference's (empty) tinylogger submodule: the render-path sources only name tlog in code we do not build.
#pragma once
#include <iostream>
#include <string>
#include <vector>
namespace tlog { // (round 5: merge_parent_network_config logs the parent's path through tlog::info(); swallowed)
struct Sink { template <class T> Sink& operator<<(const T&) { return *this; } };
inline Sink info() { return Sink(); }
inline Sink warning() { return Sink(); }
inline Sink success() { return Sink(); }
inline Sink error() { return Sink(); }
} // namespace tlog
