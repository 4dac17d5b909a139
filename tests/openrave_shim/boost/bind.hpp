// Test double for <boost/bind.hpp> (tests/openrave_shim/README.md): boost::bind and the global placeholders.
#pragma once
#include <functional>
namespace boost { using std::bind; }
using std::placeholders::_1;
using std::placeholders::_2;
