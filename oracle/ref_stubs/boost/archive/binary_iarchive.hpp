#pragma once
#include "binary_oarchive.hpp"
namespace boost { namespace archive {
class binary_iarchive {
public:
  typedef std::true_type is_loading; typedef std::false_type is_saving;
  template <class S> binary_iarchive(S &, unsigned = 0) {}
  template <class T> binary_iarchive & operator&(T &&) { throw archive_exception(); }
  template <class T> binary_iarchive & operator>>(T &&) { throw archive_exception(); }
};
}}
