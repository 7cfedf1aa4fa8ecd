/* Config::LogMode::All debug output of the extraction worker (popsift.cpp:201-209). */
#pragma once
#include "popsift/features.h"
#include "popsift/sift_conf.h"

struct popsift_hip_ctx;

namespace popsift {
namespace debug {

/* Pyramid::download_and_save_array (sift_pyramid.cu:79-83, sift_octave.cu:110-187):
 * dir-octave/, dir-octave-dump/, dir-dog/, dir-dog-txt/, dir-dog-dump/ below the current directory */
void download_and_save_array(popsift_hip_ctx* ctx, const Config& conf, const char* basename);

/* Pyramid::save_descriptors (sift_pyramid.cu:88-106): dir-desc/desc-<basename>.txt and
 * dir-fpt/desc-<basename>.txt */
void save_descriptors(const Config& conf, FeaturesHost* features, const char* basename);

}  // namespace debug
}  // namespace popsift
