// NOT Boost (see ../README.md): boost::bind and the global placeholders _1, _2 as Boost.Bind's header provides them
#pragma once
#include <functional>
namespace boost { template <class F, class... A> auto bind(F f, A... a) -> decltype(std::bind(f, a...)) { return std::bind(f, a...); } }
using std::placeholders::_1;
using std::placeholders::_2;
