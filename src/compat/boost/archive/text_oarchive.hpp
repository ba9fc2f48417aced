// apps/yelp/yelp.cpp includes <boost/archive/text_oarchive.hpp> but uses nothing from it directly: the host layer's
// own save()/load() (src/base/io/file.hpp) write a plain binary cache instead of a boost archive.
#ifndef CDAE_COMPAT_BOOST_ARCHIVE_TEXT_OARCHIVE_HPP_
#define CDAE_COMPAT_BOOST_ARCHIVE_TEXT_OARCHIVE_HPP_
#endif
