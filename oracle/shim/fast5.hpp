// TEST INFRASTRUCTURE (oracle/_ref build only) -- stand-in for mateidavid/fast5's
// <fast5.hpp>, an un-vendored submodule of the reference (/root/reference/.gitmodules).
// The reference's event_detector.hpp:10 includes it but uses nothing from it except
// the standard headers it pulled in transitively.
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
