/*
 * popsift-demo -- the reference's demo program (src/application/main.cpp:48-375) against this
 * library: same options, same console lines, same "output-features.txt".
 *
 * The reference parses its command line with boost::program_options; Boost is not a dependency of
 * this build, so the option table below carries its own small parser with the same surface:
 * `--name value`, `--name=value`, `-i value`, bool switches without a value, and unambiguous
 * prefixes of long names (program_options' default "allow_guessing" style).  Images are read by
 * the PGM/PPM reader (the DevIL path of the reference is a build option there; `--pgmread-loading`
 * is accepted and is the only loader here).
 *
 * Extensions (not in the reference): `--output-file F` (default output-features.txt),
 * `--max-extrema N`.
 */
#include <popsift/common/device_prop.h>
#include <popsift/features.h>
#include <popsift/popsift.h>
#include <popsift/sift_conf.h>

#include <chrono>
#include <cstdlib>
#include <cstring>
#include <filesystem>
#include <fstream>
#include <functional>
#include <iostream>
#include <list>
#include <queue>
#include <string>
#include <vector>

#include "cli_options.h"
#include "pgmread.h"

using namespace std;

static bool   print_dev_info = false;
static bool   print_time_info = false;
static bool   write_as_uchar = false;
static bool   dont_write = false;
static bool   pgmread_loading = false;
static bool   float_mode = false;
static string output_file = "output-features.txt";

static void parseargs(int argc, char** argv, popsift::Config& config, string& inputFile)
{
    cli::Options opts;
    auto         flag = [&](const string& n, char s, const string& g, const string& h, function<void()> f) {
        opts.flag(n, s, g, h, f);
    };
    auto val = [&](const string& n, char s, const string& g, const string& h, function<void(const string&)> f) {
        opts.val(n, s, g, h, f, n == "input-file");
    };
    auto fval = [&](const string& n, const string& g, const string& h, function<void(float)> f) { opts.fval(n, g, h, f); };
    auto ival = [&](const string& n, const string& g, const string& h, function<void(int)> f) { opts.ival(n, g, h, f); };

    /* main.cpp:51-60 */
    flag("help", 'h', "Options", "Print usage", [] {});
    flag("verbose", 'v', "Options", "", [&] { config.setVerbose(); });
    flag("log", 'l', "Options", "Write debugging files", [&] { config.setLogMode(popsift::Config::All); });
    val("input-file", 'i', "Options", "Input file", [&](const string& v) { inputFile = v; });
    /* main.cpp:61-73 */
    ival("octaves", "Parameters", "Number of octaves", [&](int v) { config.octaves = v; });
    ival("levels", "Parameters", "Number of levels per octave", [&](int v) { config.levels = v; });
    fval("sigma", "Parameters", "Initial sigma value", [&](float f) { config.setSigma(f); });
    fval("threshold", "Parameters", "Contrast threshold", [&](float f) { config.setThreshold(f); });
    fval("edge-threshold", "Parameters", "On-edge threshold", [&](float f) { config.setEdgeLimit(f); });
    fval("edge-limit", "Parameters", "On-edge threshold", [&](float f) { config.setEdgeLimit(f); });
    fval("downsampling", "Parameters", "Downscale width and height of input by 2^N",
         [&](float f) { config.setDownsampling(f); });
    fval("initial-blur", "Parameters", "Assume initial blur, subtract when blurring first time",
         [&](float f) { config.setInitialBlur(f); });
    /* main.cpp:74-112 */
    val("gauss-mode", 0, "Modes", popsift::Config::getGaussModeUsage(), [&](const string& s) { config.setGaussMode(s); });
    val("desc-mode", 0, "Modes",
        "Choice of descriptor extraction modes:\nloop, iloop, grid, igrid, notile\nDefault is loop\n"
        "loop is OpenCV-like horizontal scanning, computing only valid points, grid extracts only useful points "
        "but rounds them, iloop uses linear texture and rotated gradiant fetching. igrid is grid with linear "
        "interpolation. notile is like igrid but avoids redundant gradiant fetching.",
        [&](const string& s) { config.setDescMode(s); });
    flag("popsift-mode", 0, "Modes",
         "During the initial upscale, shift pixels by 1. In extrema refinement, steps up to 0.6, do not reject points "
         "when reaching max iterations, first contrast threshold is .8 * peak thresh. Shift feature coords octave 0 "
         "back to original pos.",
         [&] { config.setMode(popsift::Config::PopSift); });
    flag("vlfeat-mode", 0, "Modes",
         "During the initial upscale, shift pixels by 1. That creates a sharper upscaled image. In extrema refinement, "
         "steps up to 0.6, levels remain unchanged, do not reject points when reaching max iterations, first contrast "
         "threshold is .8 * peak thresh.",
         [&] { config.setMode(popsift::Config::VLFeat); });
    flag("opencv-mode", 0, "Modes",
         "During the initial upscale, shift pixels by 0.5. In extrema refinement, steps up to 0.5, reject points when "
         "reaching max iterations, first contrast threshold is floor(.5 * peak thresh). Computed filter width are "
         "lower than VLFeat/PopSift",
         [&] { config.setMode(popsift::Config::OpenCV); });
    flag("direct-scaling", 0, "Modes", "Direct each octave from upscaled orig instead of blurred level.",
         [&] { config.setScalingMode(popsift::Config::ScaleDirect); });
    ival("norm-multi", "Modes", "Multiply the descriptor by pow(2,<int>).",
         [&](int i) { config.setNormalizationMultiplier(i); });
    val("norm-mode", 0, "Modes", popsift::Config::getNormModeUsage(), [&](const string& s) { config.setNormMode(s); });
    flag("root-sift", 0, "Modes", popsift::Config::getNormModeUsage(),
         [&] { config.setNormMode(popsift::Config::RootSift); });
    ival("filter-max-extrema", "Modes", "Approximate max number of extrema.",
         [&](int f) { config.setFilterMaxExtrema(f); });
    ival("filter-grid", "Modes", "Grid edge length for extrema filtering (ie. value 4 leads to a 4x4 grid)",
         [&](int f) { config.setFilterGridSize(f); });
    val("filter-sort", 0, "Modes", "Sort extrema in each cell by scale, either random (default), up or down",
        [&](const string& s) { config.setFilterSorting(s); });
    /* main.cpp:114-126 */
    flag("print-gauss-tables", 0, "Informational", "A debug output printing Gauss filter size and tables",
         [&] { config.setPrintGaussTables(); });
    flag("print-dev-info", 0, "Informational", "A debug output printing CUDA device information",
         [&] { print_dev_info = true; });
    flag("print-time-info", 0, "Informational", "A debug output printing image processing time after load()",
         [&] { print_time_info = true; });
    flag("write-as-uchar", 0, "Informational",
         "Output descriptors rounded to int.\nScaling to sensible ranges is not automatic, should be combined with "
         "--norm-multi=9 or similar",
         [&] { write_as_uchar = true; });
    flag("dont-write", 0, "Informational", "Suppress descriptor output", [&] { dont_write = true; });
    flag("pgmread-loading", 0, "Informational", "Use the old image loader instead of LibDevIL",
         [&] { pgmread_loading = true; });
    flag("float-mode", 0, "Informational", "Upload image to GPU as float instead of byte", [&] { float_mode = true; });
    /* extensions */
    val("output-file", 0, "Extensions", "Feature file to write (default output-features.txt)",
        [&](const string& s) { output_file = s; });
    ival("max-extrema", "Extensions", "Extrema kept per octave (default 100000)", [&](int v) { config.setMaxExtrema(v); });

    opts.parse(argc, argv);
}

static void collectFilenames(list<string>& inputFiles, const filesystem::path& inputFile)
{
    /* main.cpp:149-166: regular files of the directory, sub-directories recursively */
    vector<filesystem::path> vec;
    for (const auto& entry : filesystem::directory_iterator(inputFile)) vec.push_back(entry.path());
    for (const auto& p : vec) {
        if (filesystem::is_regular_file(p)) inputFiles.push_back(p.string());
        else if (filesystem::is_directory(p)) collectFilenames(inputFiles, p);
    }
}

static SiftJob* process_image(const string& inputFile, PopSift& sift)
{
    /* main.cpp:168-247, pgmread branch */
    int            w, h;
    unsigned char* image_data = readPGMfile(inputFile, w, h);
    if (image_data == 0) exit(-1);
    SiftJob* job;
    if (!float_mode) {
        job = sift.enqueue(w, h, image_data);
    } else {
        float* f_image_data = new float[(size_t)w * h];
        for (size_t i = 0; i < (size_t)w * h; i++) f_image_data[i] = float(image_data[i]) / 256.0f;
        job = sift.enqueue(w, h, f_image_data);
        delete[] f_image_data;
    }
    delete[] image_data;
    return job;
}

static void read_job(SiftJob* job, bool really_write)
{
    /* main.cpp:249-267: every job rewrites the same file, the last image wins */
    popsift::Features* feature_list = job->get();
    cerr << "Number of feature points: " << feature_list->getFeatureCount()
         << " number of feature descriptors: " << feature_list->getDescriptorCount() << endl;
    if (really_write) {
        std::ofstream of(output_file.c_str());
        feature_list->print(of, write_as_uchar);
    }
    delete feature_list;
}

int main(int argc, char** argv)
{
    popsift::Config config;
    list<string>    inputFiles;
    string          inputFile = "";

    parseargs(argc, argv, config, inputFile);
    std::cout << inputFile << std::endl;

    if (filesystem::exists(inputFile)) {
        if (filesystem::is_directory(inputFile)) {
            cout << "BOOST " << inputFile << " is directory" << endl;
            collectFilenames(inputFiles, inputFile);
            if (inputFiles.empty()) {
                cerr << "No files in directory, nothing to do" << endl;
                exit(0);
            }
        } else if (filesystem::is_regular_file(inputFile)) {
            inputFiles.push_back(inputFile);
        } else {
            cout << "Input file is neither regular file nor directory, nothing to do" << endl;
            exit(-1);
        }
    }

    popsift::cuda::device_prop_t deviceInfo;
    deviceInfo.set(0, print_dev_info);
    if (print_dev_info) deviceInfo.print();

    PopSift sift(config, popsift::Config::ExtractingMode, float_mode ? PopSift::FloatImages : PopSift::ByteImages);

    const auto           t0 = chrono::steady_clock::now();
    std::queue<SiftJob*> jobs;
    for (const string& name : inputFiles) jobs.push(process_image(name, sift));

    while (!jobs.empty()) {
        SiftJob* job = jobs.front();
        jobs.pop();
        if (job) {
            read_job(job, !dont_write);
            delete job;
        }
    }
    if (print_time_info) {
        const double ms = chrono::duration<double, milli>(chrono::steady_clock::now() - t0).count();
        cerr << "Time for " << inputFiles.size() << " image(s), load to features: " << ms << " ms" << endl;
    }
    sift.uninit();
    return 0;
}
