// Stand-in for Boost.Serialization used ONLY to compile the reference karto_sdk
// sources as a test oracle (oracle/_ref). Serialization itself is never executed.
#pragma once
#include <cstddef>
#define BOOST_SERIALIZATION_NVP(x) x
#define BOOST_SERIALIZATION_BASE_OBJECT_NVP(T) (*static_cast<T *>(this))
#define BOOST_SERIALIZATION_ASSUME_ABSTRACT(T)
#define BOOST_CLASS_EXPORT(T)
#define BOOST_CLASS_EXPORT_KEY(T) static_assert(true, "")
#define BOOST_CLASS_EXPORT_IMPLEMENT(T) static_assert(true, "")
namespace boost { namespace serialization {
class access {};
template <class T> inline T & make_nvp(const char *, T & t) { return t; }
template <class T> struct array_ref { T * p; std::size_t n; };
template <class T> inline array_ref<T> make_array(T * p, std::size_t n) { return array_ref<T>{p, n}; }
template <class B, class D> inline B & base_object(D & d) { return static_cast<B &>(d); }
}}
