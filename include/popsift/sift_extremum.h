/*
 * popsift/sift_extremum.h -- the 128-float SIFT descriptor as handed to callers.
 * Replaces sift_extremum.h:56-59 (the device-internal InitialExtremum / Extremum
 * records of that file are private to libpopsift_hip).
 */
#pragma once

#include "sift_constants.h"

namespace popsift {

struct Descriptor {
    float features[128];
};

}  // namespace popsift
