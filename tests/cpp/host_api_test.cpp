// Host-side checks of the drop-in C++ API that need no GPU: Config defaults / setters / equal(),
// FeaturesHost storage and the text format of Feature::print (features.cu:308-328).
#include <popsift/features.h>
#include <popsift/popsift.h>
#include <popsift/sift_conf.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <sstream>
#include <string>
#include <thread>
#include <vector>

#define CHECK(c)                                                         \
    do {                                                                 \
        if (!(c)) {                                                      \
            std::fprintf(stderr, "CHECK failed %s:%d: %s\n", __FILE__, __LINE__, #c); \
            std::exit(1);                                                \
        }                                                                \
    } while (0)

int main()
{
    popart::Config alias;  // namespace alias of the README (popart::Config)
    popsift::Config c;
    // defaults of sift_conf.cu:17-39
    CHECK(c.octaves == -1 && c.levels == 3 && c.sigma == 1.6f && c._edge_limit == 10.0f && !c.verbose);
    CHECK(c.getUpscaleFactor() == 1.0f && c.getMaxExtrema() == 100000 && c.getFilterMaxExtrema() == -1);
    CHECK(c.getFilterGridSize() == 2 && c.hasInitialBlur() && c.getInitialBlur() == 0.5f);
    CHECK(c.getGaussMode() == popsift::Config::VLFeat_Compute && c.getSiftMode() == popsift::Config::PopSift);
    CHECK(c.getDescMode() == popsift::Config::Loop && c.getUseRootSift() && c.getNormalizationMultiplier() == 0);
    CHECK(std::fabs(c.getPeakThreshold() - 1.7f) < 1e-6f);   // 0.04 * 0.5 * 255 / 3
    CHECK(c.getLogMode() == popsift::Config::None && c.getScalingMode() == popsift::Config::ScaleDefault);
    CHECK(popsift::Config::Default == popsift::Config::PopSift);
    CHECK(c == alias);

    // setters
    c.setDownsampling(-1.0f);
    CHECK(c.getUpscaleFactor() == 1.0f);   // stores the negation (sift_conf.cu:234)
    c.setDownsampling(1.0f);
    CHECK(c.getUpscaleFactor() == -1.0f && c != alias);
    c.setDownsampling(-1.0f);
    CHECK(c == alias);
    c.setGaussMode("opencv");
    CHECK(c.getGaussMode() == popsift::Config::OpenCV_Compute && c != alias);
    c.setGaussMode("vlfeat");
    c.setDescMode("loop");
    c.setNormMode("classic");
    CHECK(!c.getUseRootSift() && c != alias);
    c.setNormMode("RootSift");
    c.setInitialBlur(0.0f);
    CHECK(!c.hasInitialBlur());
    c.setInitialBlur(0.5f);
    CHECK(c.hasInitialBlur() && c == alias);
    c.setLevels(4);
    CHECK(std::fabs(c.getPeakThreshold() - 0.04f * 0.5f * 255.0f / 4) < 1e-6f && c != alias);
    c.setLevels(3);
    c.setFilterSorting("up");
    CHECK(c.getFilterSorting() == popsift::Config::SmallestScaleFirst);
    CHECK(c == alias);   // the grid-filter fields are not part of equal()
    c.setMode(popsift::Config::VLFeat);
    CHECK(c != alias);

    // result containers
    popsift::FeaturesHost fh(3, 4);
    CHECK(fh.size() == 3 && fh.getFeatureCount() == 3 && fh.getDescriptorCount() == 4);
    CHECK(fh.end() - fh.begin() == 3 && ((size_t)fh.getDescriptors() % 4096) == 0);
    CHECK(sizeof(popsift::Descriptor) == 512 && sizeof(popsift::Feature) == 72 && ORIENTATION_MAX_COUNT == 4);
    popsift::Feature& f = fh.getFeatures()[0];
    f.xpos = 10.5f; f.ypos = 20.25f; f.sigma = 2.0f; f.num_ori = 1; f.debug_octave = 0;
    f.desc[0] = fh.getDescriptors() + 2;
    for (int i = 0; i < 128; i++) f.desc[0]->features[i] = 0.125f;
    std::ostringstream os;
    f.print(os, false);
    const std::string line = os.str();
    CHECK(line.rfind("10.5 20.25 0.25 0 0.25 0.125 ", 0) == 0);   // x y 1/s^2 0 1/s^2 d0 ...
    popsift::FeaturesHost empty(0, 0);
    CHECK(empty.size() == 0 && empty.begin() == empty.end());
    // result blocks come from a pool: a freed result is cached for reuse, releasePinnedCache() returns it to the system
    {
        popsift::FeaturesHost* big = new popsift::FeaturesHost(1000, 3000);
        popsift::Descriptor*   where = big->getDescriptors();
        delete big;
        CHECK(popsift::pinnedCacheBytes() >= 3000 * sizeof(popsift::Descriptor));
        popsift::FeaturesHost again(900, 2900);          // a little smaller: same size class, same block
        CHECK(again.getDescriptors() == where);
    }
    popsift::releasePinnedCache();
    CHECK(popsift::pinnedCacheBytes() == 0);

    // a SiftJob keeps its copy of the image in a block of the same pool: the caller's buffer is free at once, and the
    // block of a deleted job serves the next one (no allocation per job)
    {
        std::vector<unsigned char> img(640 * 480, 7);
        SiftJob*                   j = new SiftJob(640, 480, img.data());
        const unsigned char*       where = j->getImageData();
        img.assign(img.size(), 9);
        CHECK(where != img.data() && where[0] == 7 && where[640 * 480 - 1] == 7 && ((size_t)where % 4096) == 0);
        delete j;
        SiftJob k(640, 480, img.data());
        CHECK(k.getImageData() == where && k.getImageData()[5] == 9);
        std::vector<float> fimg(64 * 48, 0.5f);
        SiftJob            jf(64, 48, fimg.data());
        CHECK(jf.isFloat() && ((const float*)jf.getImageData())[64 * 48 - 1] == 0.5f);
    }
    popsift::releasePinnedCache();

    // the sysfs cpulist parser behind the workers' NUMA binding, from several threads at once (an earlier version used
    // strtok, whose single process-wide state made concurrent workers corrupt each other's parse)
    {
        std::vector<std::thread> th;
        std::vector<int>         bad(8, 0);
        for (int t = 0; t < 8; t++)
            th.emplace_back([t, &bad] {
                for (int it = 0; it < 2000; it++) {
                    cpu_set_t set;
                    char      list[64];
                    std::snprintf(list, sizeof(list), "%d-%d,%d-%d,%d\n", t, t + 3, 64 + t, 64 + t + 7, 200 + t);
                    const int n = popsift::parseCpuList(list, &set);
                    bool      ok = (n == 13) && CPU_COUNT(&set) == 13;
                    for (int c = t; c <= t + 3; c++) ok = ok && CPU_ISSET(c, &set);
                    for (int c = 64 + t; c <= 64 + t + 7; c++) ok = ok && CPU_ISSET(c, &set);
                    ok = ok && CPU_ISSET(200 + t, &set) && !CPU_ISSET(t + 4, &set);
                    if (!ok) bad[t]++;
                }
            });
        for (auto& x : th) x.join();
        for (int t = 0; t < 8; t++) CHECK(bad[t] == 0);
        cpu_set_t set;
        CHECK(popsift::parseCpuList("", &set) == 0 && popsift::parseCpuList("\n", &set) == 0);
        CHECK(popsift::parseCpuList("0-3,2-5", &set) == 6 && CPU_COUNT(&set) == 6);   // overlapping ranges count once
        CHECK(popsift::parseCpuList("7", &set) == 1 && CPU_ISSET(7, &set));
        CHECK(popsift::parseCpuList("0-1,junk", &set) == 2);                            // stops at what it understands
    }

    // a PopSift object can be made and torn down without ever touching a GPU
    {
        PopSift p(PopSift::ByteImages);
        CHECK(p.configure(popsift::Config()));
        CHECK(p.getContextCount() == 0);
        p.uninit();
    }
    std::printf("host_api_test ok\n");
    return 0;
}
