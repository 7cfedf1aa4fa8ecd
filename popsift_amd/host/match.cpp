/*
 * popsift-match -- the reference's matching demo (src/application/match.cpp:47-276) against this
 * library: extract two images in Config::MatchingMode (features stay on the GPU), print the counts,
 * brute-force match left against right (one accept / reject line per left descriptor).
 */
#include <popsift/common/device_prop.h>
#include <popsift/features.h>
#include <popsift/popsift.h>
#include <popsift/sift_conf.h>

#include <cstdlib>
#include <filesystem>
#include <functional>
#include <iostream>
#include <string>

#include "cli_options.h"
#include "pgmread.h"

using namespace std;

static bool print_dev_info = false;
static bool print_time_info = false;
static bool write_as_uchar = false;
static bool dont_write = false;
static bool pgmread_loading = false;

static void parseargs(int argc, char** argv, popsift::Config& config, string& lFile, string& rFile)
{
    cli::Options o;
    /* match.cpp:50-60 */
    o.flag("help", 'h', "Options", "Print usage", [] {});
    o.flag("verbose", 'v', "Options", "", [&] { config.setVerbose(); });
    o.flag("log", 0, "Options", "Write debugging files", [&] { config.setLogMode(popsift::Config::All); });
    o.val("left", 'l', "Options", "\"Left\"  input file", [&](const string& v) { lFile = v; }, true);
    o.val("right", 'r', "Options", "\"Right\" input file", [&](const string& v) { rFile = v; }, true);
    /* match.cpp:61-73 */
    o.ival("octaves", "Parameters", "Number of octaves", [&](int v) { config.octaves = v; });
    o.ival("levels", "Parameters", "Number of levels per octave", [&](int v) { config.levels = v; });
    o.fval("sigma", "Parameters", "Initial sigma value", [&](float f) { config.setSigma(f); });
    o.fval("threshold", "Parameters", "Contrast threshold", [&](float f) { config.setThreshold(f); });
    o.fval("edge-threshold", "Parameters", "On-edge threshold", [&](float f) { config.setEdgeLimit(f); });
    o.fval("edge-limit", "Parameters", "On-edge threshold", [&](float f) { config.setEdgeLimit(f); });
    o.fval("downsampling", "Parameters", "Downscale width and height of input by 2^N",
           [&](float f) { config.setDownsampling(f); });
    o.fval("initial-blur", "Parameters", "Assume initial blur, subtract when blurring first time",
           [&](float f) { config.setInitialBlur(f); });
    /* match.cpp:74-112 */
    o.val("gauss-mode", 0, "Modes", popsift::Config::getGaussModeUsage(), [&](const string& s) { config.setGaussMode(s); });
    o.val("desc-mode", 0, "Modes", "Choice of descriptor extraction modes:\nloop, iloop, grid, igrid, notile\nDefault is loop",
          [&](const string& s) { config.setDescMode(s); });
    o.flag("popsift-mode", 0, "Modes", "PopSift's own refinement and upscale rules (default)",
           [&] { config.setMode(popsift::Config::PopSift); });
    o.flag("vlfeat-mode", 0, "Modes", "VLFeat-like refinement (levels stay fixed)", [&] { config.setMode(popsift::Config::VLFeat); });
    o.flag("opencv-mode", 0, "Modes", "OpenCV-like upscale shift, refinement steps and filter widths",
           [&] { config.setMode(popsift::Config::OpenCV); });
    o.flag("direct-scaling", 0, "Modes", "Direct each octave from upscaled orig instead of blurred level.",
           [&] { config.setScalingMode(popsift::Config::ScaleDirect); });
    o.ival("norm-multi", "Modes", "Multiply the descriptor by pow(2,<int>).", [&](int i) { config.setNormalizationMultiplier(i); });
    o.val("norm-mode", 0, "Modes", popsift::Config::getNormModeUsage(), [&](const string& s) { config.setNormMode(s); });
    o.flag("root-sift", 0, "Modes", popsift::Config::getNormModeUsage(), [&] { config.setNormMode(popsift::Config::RootSift); });
    o.ival("filter-max-extrema", "Modes", "Approximate max number of extrema.", [&](int f) { config.setFilterMaxExtrema(f); });
    o.ival("filter-grid", "Modes", "Grid edge length for extrema filtering (ie. value 4 leads to a 4x4 grid)",
           [&](int f) { config.setFilterGridSize(f); });
    o.val("filter-sort", 0, "Modes", "Sort extrema in each cell by scale, either random (default), up or down",
          [&](const string& s) { config.setFilterSorting(s); });
    /* match.cpp:114-125 */
    o.flag("print-gauss-tables", 0, "Informational", "A debug output printing Gauss filter size and tables",
           [&] { config.setPrintGaussTables(); });
    o.flag("print-dev-info", 0, "Informational", "A debug output printing CUDA device information", [&] { print_dev_info = true; });
    o.flag("print-time-info", 0, "Informational", "A debug output printing image processing time after load()",
           [&] { print_time_info = true; });
    o.flag("write-as-uchar", 0, "Informational", "Output descriptors rounded to int", [&] { write_as_uchar = true; });
    o.flag("dont-write", 0, "Informational", "Suppress descriptor output", [&] { dont_write = true; });
    o.flag("pgmread-loading", 0, "Informational", "Use the old image loader instead of LibDevIL", [&] { pgmread_loading = true; });
    o.parse(argc, argv);
}

static SiftJob* process_image(const string& inputFile, PopSift& sift)
{
    int            w, h;
    unsigned char* image_data = readPGMfile(inputFile, w, h);
    if (image_data == 0) exit(-1);
    SiftJob* job = sift.enqueue(w, h, image_data);
    delete[] image_data;
    return job;
}

int main(int argc, char** argv)
{
    popsift::Config config;
    string          lFile = "";
    string          rFile = "";

    parseargs(argc, argv, config, lFile, rFile);
    std::cout << lFile << " <-> " << rFile << std::endl;

    for (const string& f : {lFile, rFile}) {
        if (filesystem::exists(f) && !filesystem::is_regular_file(f)) {
            cout << "Input file " << f << " is not a regular file, nothing to do" << endl;
            exit(-1);
        }
    }

    popsift::cuda::device_prop_t deviceInfo;
    deviceInfo.set(0, print_dev_info);
    if (print_dev_info) deviceInfo.print();

    PopSift sift(config, popsift::Config::MatchingMode);

    SiftJob* lJob = process_image(lFile, sift);
    SiftJob* rJob = process_image(rFile, sift);

    popsift::FeaturesDev* lFeatures = lJob->getDev();
    cout << "Number of features:    " << lFeatures->getFeatureCount() << endl;
    cout << "Number of descriptors: " << lFeatures->getDescriptorCount() << endl;

    popsift::FeaturesDev* rFeatures = rJob->getDev();
    cout << "Number of features:    " << rFeatures->getFeatureCount() << endl;
    cout << "Number of descriptors: " << rFeatures->getDescriptorCount() << endl;

    lFeatures->match(rFeatures);

    delete lFeatures;
    delete rFeatures;
    delete lJob;
    delete rJob;

    sift.uninit();
    return 0;
}
