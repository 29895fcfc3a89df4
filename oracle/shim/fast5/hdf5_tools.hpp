// TEST INFRASTRUCTURE (oracle/_ref build only) -- stand-in for fast5/hdf5_tools.hpp.
// Only the interface that read_buffer.cpp:198-246 names is declared; the oracle harness
// never opens an HDF5 file (signals are handed over as arrays), so every method is a stub.
#pragma once
#include <array>
#include <cstring>
#include <deque>
#include <fstream>
#include <iostream>
#include <map>
#include <mutex>
#include <string>
#include <vector>
namespace hdf5_tools {
struct File {
    bool is_open() const { return false; }
    void open(const std::string &) {}
    void close() {}
    std::vector<std::string> list_group(const std::string &) const { return {}; }
    std::map<std::string, std::string> get_attr_map(const std::string &) const { return {}; }
    template <class T> void read(const std::string &, std::vector<T> &) const {}
};
}  // namespace hdf5_tools
