// IO umbrella (reference: src/base/io.hpp).
#ifndef CDAE_HOST_BASE_IO_HPP_
#define CDAE_HOST_BASE_IO_HPP_
#include <base/io/file.hpp>
#endif
