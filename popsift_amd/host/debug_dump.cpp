/*
 * Debug output for Config::setLogMode(All) (the demo's --log): same directories, file names
 * and file formats as the reference, so that tools written against its dumps keep working.
 *   planes:      sift_octave.cu:110-187 + common/write_plane_2d.cu:50-175
 *   descriptors: sift_pyramid.cu:88-106, 400-448
 */
#include "debug_dump.h"

#include <cmath>
#include <fstream>
#include <iomanip>
#include <limits>
#include <sstream>
#include <sys/stat.h>
#include <vector>

#include "popsift_hip.h"

using namespace std;

namespace popsift {
namespace debug {

namespace {

void make_dir(const char* name)
{
    struct stat st;
    if (stat(name, &st) == -1) mkdir(name, 0700);
}

/* write_plane2D (write_plane_2d.cu:50-107): values stretched to 0..255, plain PGM.  The maximum
 * search starts from numeric_limits<float>::min() -- the smallest POSITIVE float -- as there. */
void write_scaled(const string& name, const float* p, int cols, int rows)
{
    float minval = numeric_limits<float>::max();
    float maxval = numeric_limits<float>::min();
    for (int i = 0; i < rows * cols; i++) {
        minval = min(minval, p[i]);
        maxval = max(maxval, p[i]);
    }
    const float fmaxval = 255.0f / (maxval - minval);
    ofstream    of(name.c_str(), ios::binary);
    of << "P2" << endl << cols << " " << rows << endl << "255" << endl;
    for (int y = 0; y < rows; y++) {
        for (int x = 0; x < cols; x++) {
            const unsigned char c = (unsigned char)((p[y * cols + x] - minval) * fmaxval);
            of << (int)c << " ";
        }
        of << endl;
    }
}

/* write_plane2Dunscaled (write_plane_2d.cu:110-139): values truncated to int, plus an offset */
void write_unscaled(const string& name, const float* p, int cols, int rows, int offset)
{
    ofstream of(name.c_str(), ios::binary);
    of << "P2" << endl << cols << " " << rows << endl << "255" << endl;
    for (int y = 0; y < rows; y++) {
        for (int x = 0; x < cols; x++) of << (int)p[y * cols + x] + offset << " ";
        of << endl;
    }
}

/* dump_plane2Dfloat (write_plane_2d.cu:157-175): "floats", the size, the raw floats */
void dump_floats(const string& name, const float* p, int cols, int rows)
{
    ofstream of(name.c_str(), ios::binary);
    of << "floats" << endl << cols << " " << rows << endl;
    of.write((const char*)p, (streamsize)((size_t)rows * cols * sizeof(float)));
}

string plane_name(const char* dir, const char* prefix, const char* basename, int octave, int level, const char* ext)
{
    ostringstream o;
    o << dir << "/" << prefix << basename << "-o-" << octave << "-l-" << level << ext;
    return o.str();
}

/* Pyramid::writeDescriptor (sift_pyramid.cu:400-448).  The positions in `features` are already in
 * input-image coordinates (prep_features) and are scaled by 2^(octave - upscale) AGAIN here, as in
 * the reference: this is debug output and tools compare it verbatim. */
void write_descriptors(const Config& conf, ostream& ostr, FeaturesHost* features, bool really, bool with_orientation)
{
    if (features->getFeatureCount() == 0) return;
    const float up_fac = conf.getUpscaleFactor();
    const float two_pi = 2.0f * 3.14159265358979323846f; /* M_PI2, sift_constants.h */
    for (int i = 0; i < features->getFeatureCount(); i++) {
        const Feature& ext = features->getFeatures()[i];
        const int      octave = ext.debug_octave;
        const float    xpos = ext.xpos * pow(2.0f, octave - up_fac);
        const float    ypos = ext.ypos * pow(2.0f, octave - up_fac);
        const float    sigma = ext.sigma * pow(2.0f, octave - up_fac);
        for (int ori = 0; ori < ext.num_ori; ori++) {
            float dom_ori = ext.orientation[ori];
            dom_ori = dom_ori / two_pi * 360;
            if (dom_ori < 0) dom_ori += 360;
            if (with_orientation)
                ostr << setprecision(5) << xpos << " " << ypos << " " << sigma << " " << dom_ori << " ";
            else
                ostr << setprecision(5) << xpos << " " << ypos << " " << 1.0f / (sigma * sigma) << " 0 "
                     << 1.0f / (sigma * sigma) << " ";
            if (really)
                for (int k = 0; k < 128; k++) ostr << ext.desc[ori]->features[k] << " ";
            ostr << endl;
        }
    }
}

}  // namespace

void download_and_save_array(popsift_hip_ctx* ctx, const Config& conf, const char* basename)
{
    make_dir("dir-octave");
    make_dir("dir-octave-dump");
    make_dir("dir-dog");
    make_dir("dir-dog-txt");
    make_dir("dir-dog-dump");
    const int levels = max(2, conf.levels) + 3;
    for (int o = 0;; o++) {
        int w = 0, h = 0;
        if (popsift_hip_octave_dims(ctx, o, &w, &h) != POPSIFT_HIP_OK) break;
        vector<float> plane((size_t)w * h);
        for (int l = 0; l < levels; l++) {
            if (popsift_hip_download_plane(ctx, o, 0, l, plane.data()) != POPSIFT_HIP_OK) continue;
            write_unscaled(plane_name("dir-octave", "", basename, o, l, ".pgm"), plane.data(), w, h, 0);
            dump_floats(plane_name("dir-octave-dump", "", basename, o, l, ".dump"), plane.data(), w, h);
        }
        for (int l = 0; l < levels - 1; l++) {
            if (popsift_hip_download_plane(ctx, o, 1, l, plane.data()) != POPSIFT_HIP_OK) continue;
            write_scaled(plane_name("dir-dog", "d-", basename, o, l, ".pgm"), plane.data(), w, h);
            write_unscaled(plane_name("dir-dog-txt", "d-", basename, o, l, ".txt"), plane.data(), w, h, 127);
            dump_floats(plane_name("dir-dog-dump", "d-", basename, o, l, ".dump"), plane.data(), w, h);
        }
    }
}

void save_descriptors(const Config& conf, FeaturesHost* features, const char* basename)
{
    make_dir("dir-desc");
    {
        ostringstream n;
        n << "dir-desc/desc-" << basename << ".txt";
        ofstream of(n.str().c_str());
        write_descriptors(conf, of, features, true, true);
    }
    make_dir("dir-fpt");
    {
        ostringstream n;
        n << "dir-fpt/desc-" << basename << ".txt";
        ofstream of(n.str().c_str());
        write_descriptors(conf, of, features, false, true);
    }
}

}  // namespace debug
}  // namespace popsift
