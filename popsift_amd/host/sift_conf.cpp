/* popsift::Config -- defaults and string setters (replaces sift_conf.cu:17-303). */
#include "popsift/sift_conf.h"

#include <cstdlib>
#include <iostream>

namespace popsift {

namespace {
[[noreturn]] void fatal(const std::string& msg)
{
    /* the reference's POP_FATAL prints and exits (common/debug_macros.h:141-146) */
    std::cerr << __FILE__ << ": " << msg << std::endl;
    std::exit(-1);
}
}  // namespace

/* sift_conf.cu:17-39 */
Config::Config()
    : octaves(-1)
    , levels(3)
    , sigma(1.6f)
    , _edge_limit(10.0f)
    , verbose(false)
    , _threshold(0.04f)
    , _upscale_factor(1.0f)
    , _log_mode(Config::None)
    , _scaling_mode(Config::ScaleDefault)
    , _desc_mode(Config::Loop)
    , _grid_filter_mode(Config::RandomScale)
    , _max_extrema(100000)
    , _filter_max_extrema(-1)
    , _filter_grid_size(2)
    , _gauss_mode(getGaussModeDefault())
    , _sift_mode(Config::PopSift)
    , _assume_initial_blur(true)
    , _initial_blur(0.5f)
    , _normalization_mode(getNormModeDefault())
    , _normalization_multiplier(0)
    , _print_gauss_tables(false)
{
}

void Config::setDescMode(const std::string& text)
{
    if (text == "loop") setDescMode(Config::Loop);
    else if (text == "iloop") setDescMode(Config::ILoop);
    else if (text == "grid") setDescMode(Config::Grid);
    else if (text == "igrid") setDescMode(Config::IGrid);
    else if (text == "notile") setDescMode(Config::NoTile);
    else fatal("specified descriptor extraction mode must be one of loop, grid or igrid");
}

void Config::setGaussMode(const std::string& m)
{
    if (m == "vlfeat") setGaussMode(Config::VLFeat_Compute);
    else if (m == "vlfeat-hw-interpolated" || m == "relative") setGaussMode(Config::VLFeat_Relative);
    else if (m == "vlfeat-direct") setGaussMode(Config::VLFeat_Relative_All);
    else if (m == "opencv") setGaussMode(Config::OpenCV_Compute);
    else if (m == "fixed9") setGaussMode(Config::Fixed9);
    else if (m == "fixed15") setGaussMode(Config::Fixed15);
    else fatal(std::string("Bad Gauss mode.\n") + getGaussModeUsage());
}

const char* Config::getGaussModeUsage()
{
    return "Choice of Gauss filter method. Options are: vlfeat (default), vlfeat-hw-interpolated, "
           "vlfeat-direct, opencv, fixed9, fixed15, relative (synonym for vlfeat-hw-interpolated)";
}

void Config::setFilterSorting(const std::string& text)
{
    if (text == "up") _grid_filter_mode = Config::SmallestScaleFirst;
    else if (text == "down") _grid_filter_mode = Config::LargestScaleFirst;
    else if (text == "random") _grid_filter_mode = Config::RandomScale;
    else fatal("filter sorting mode must be one of up, down or random");
}

void Config::setUseRootSift(bool on) { _normalization_mode = on ? RootSift : Classic; }

void Config::setNormMode(const std::string& m)
{
    if (m == "RootSift") setNormMode(Config::RootSift);
    else if (m == "classic") setNormMode(Config::Classic);
    else fatal(std::string("Bad Normalization mode.\n") + getNormModeUsage());
}

const char* Config::getNormModeUsage()
{
    return "Choice of descriptor normalization modes. Options are: RootSift (L1-like, default), Classic (L2-like)";
}

void Config::setInitialBlur(float blur)
{
    _assume_initial_blur = (blur != 0.0f);
    _initial_blur = blur;
}

bool Config::equal(const Config& o) const
{
    return octaves == o.octaves && levels == o.levels && sigma == o.sigma && _edge_limit == o._edge_limit &&
           _threshold == o._threshold && _upscale_factor == o._upscale_factor && _scaling_mode == o._scaling_mode &&
           _max_extrema == o._max_extrema && _gauss_mode == o._gauss_mode && _sift_mode == o._sift_mode &&
           _assume_initial_blur == o._assume_initial_blur && _initial_blur == o._initial_blur &&
           _normalization_mode == o._normalization_mode && _normalization_multiplier == o._normalization_multiplier;
}

}  // namespace popsift
