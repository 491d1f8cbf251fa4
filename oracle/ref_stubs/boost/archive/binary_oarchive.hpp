// Throwing stand-in archive (oracle build only; never used at run time).
#pragma once
#include <ios>
#include <stdexcept>
#include <type_traits>
namespace boost { namespace archive {
enum archive_flags { no_codecvt = 4 };
class archive_exception : public std::runtime_error {
public: archive_exception() : std::runtime_error("boost archive stub") {} };
class binary_oarchive {
public:
  typedef std::false_type is_loading; typedef std::true_type is_saving;
  template <class S> binary_oarchive(S &, unsigned = 0) {}
  template <class T> binary_oarchive & operator&(const T &) { throw archive_exception(); }
  template <class T> binary_oarchive & operator<<(const T &) { throw archive_exception(); }
};
}}
