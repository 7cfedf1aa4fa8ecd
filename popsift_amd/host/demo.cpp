/*
 * popsift-demo -- minimal caller of the drop-in API, written the way the reference's demo
 * uses it (src/application/main.cpp:304-326): construct, enqueue everything, then get()
 * every job, write "output-features.txt"-style text.  Input: binary PGM (P5) files.
 *   popsift-demo [--mode popsift|vlfeat|opencv] [--norm-mode RootSift|classic] [--octaves N]
 *                [--levels N] [--downsampling V] [--float] [-o out.txt] image.pgm ...
 */
#include <popsift/features.h>
#include <popsift/popsift.h>
#include <popsift/sift_conf.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <string>
#include <vector>

static bool read_pgm(const std::string& name, int& w, int& h, std::vector<unsigned char>& px)
{
    std::ifstream f(name.c_str(), std::ios::binary);
    if (!f) return false;
    std::string magic;
    f >> magic;
    if (magic != "P5") return false;
    int  vals[3], n = 0;
    while (n < 3 && f) {
        f >> std::ws;
        if (f.peek() == '#') {
            std::string line;
            std::getline(f, line);
            continue;
        }
        f >> vals[n++];
    }
    if (n < 3 || vals[2] > 255) return false;
    f.get();
    w = vals[0];
    h = vals[1];
    px.resize((size_t)w * h);
    f.read((char*)px.data(), (std::streamsize)px.size());
    return (bool)f;
}

int main(int argc, char** argv)
{
    popsift::Config          config;
    std::vector<std::string> files;
    std::string              out = "output-features.txt";
    bool                     float_mode = false;
    for (int i = 1; i < argc; i++) {
        const std::string a = argv[i];
        auto              next = [&]() -> const char* {
            if (i + 1 >= argc) {
                std::cerr << "missing value for " << a << std::endl;
                exit(1);
            }
            return argv[++i];
        };
        if (a == "--mode") {
            const std::string m = next();
            config.setMode(m == "vlfeat" ? popsift::Config::VLFeat : m == "opencv" ? popsift::Config::OpenCV
                                                                                    : popsift::Config::PopSift);
        } else if (a == "--gauss-mode") config.setGaussMode(next());
        else if (a == "--desc-mode") config.setDescMode(next());
        else if (a == "--norm-mode") config.setNormMode(next());
        else if (a == "--norm-multi") config.setNormalizationMultiplier(atoi(next()));
        else if (a == "--octaves") config.setOctaves(atoi(next()));
        else if (a == "--levels") config.setLevels(atoi(next()));
        else if (a == "--sigma") config.setSigma((float)atof(next()));
        else if (a == "--threshold") config.setThreshold((float)atof(next()));
        else if (a == "--edge-threshold") config.setEdgeLimit((float)atof(next()));
        else if (a == "--downsampling") config.setDownsampling((float)atof(next()));
        else if (a == "--initial-blur") config.setInitialBlur((float)atof(next()));
        else if (a == "--float") float_mode = true;
        else if (a == "-o") out = next();
        else files.push_back(a);
    }
    if (files.empty()) {
        std::cerr << "usage: popsift-demo [options] image.pgm ..." << std::endl;
        return 1;
    }

    PopSift sift(config, popsift::Config::ExtractingMode, float_mode ? PopSift::FloatImages : PopSift::ByteImages);

    std::vector<SiftJob*> jobs;
    for (const std::string& name : files) {
        int                        w, h;
        std::vector<unsigned char> px;
        if (!read_pgm(name, w, h, px)) {
            std::cerr << "cannot read " << name << " (binary PGM expected)" << std::endl;
            return 1;
        }
        if (float_mode) {
            std::vector<float> fp(px.size());
            for (size_t k = 0; k < px.size(); k++) fp[k] = float(px[k]) / 256.0f; /* main.cpp:233 */
            jobs.push_back(sift.enqueue(w, h, fp.data()));
        } else {
            jobs.push_back(sift.enqueue(w, h, px.data()));
        }
    }
    std::ofstream of(out.c_str());
    for (SiftJob* job : jobs) {
        popsift::Features* feature_list = job->get();
        std::cerr << "Number of feature points: " << feature_list->getFeatureCount()
                  << " number of feature descriptors: " << feature_list->getDescriptorCount() << std::endl;
        feature_list->print(of, false);
        delete feature_list;
        delete job;
    }
    sift.uninit();
    return 0;
}
