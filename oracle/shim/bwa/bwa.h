// TEST INFRASTRUCTURE (oracle/_ref build only) -- declarations of the slice of lh3/bwa
// (0.7.17 per example/README.md:12; un-vendored submodule) that bwa_index.hpp touches.
// The definitions live in oracle/minibwa.c, written from the published on-disk format.
#pragma once
#include "../../minibwa.h"
