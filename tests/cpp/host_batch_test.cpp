// GPU test driver for BASELINE.json configs 4 and 5 through the drop-in C++ API (PopSift::enqueue ... SiftJob::get,
// the usage pattern of src/application/main.cpp:304-326): every PGM on the command line is enqueued into ONE PopSift
// object (its worker pool spans POPSIFT_DEVICES x POPSIFT_CONTEXTS_PER_DEVICE contexts), then every job is collected.
// Per image it prints "idx features descriptors digest" -- the digest is an order-independent 64-bit sum of FNV-1a
// hashes over (x, y, sigma, num_ori, k, orientation[k], the 128 descriptor words) of every orientation, which the
// Python side recomputes from a single-context C-ABI run of the same image -- and dumps the full result of the
// images named by --dump for the comparison with the oracle.
//   host_batch_test <out_dir> [--opencv] [--dump i,j,k] a.pgm b.pgm ...
#include <popsift/features.h>
#include <popsift/popsift.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <set>
#include <sstream>
#include <string>
#include <vector>

#include "../../popsift_amd/host/pgmread.h"

static uint64_t fnv(uint64_t h, uint32_t w)
{
    h ^= w;
    return h * 1099511628211ull;
}
static uint32_t bits(float f)
{
    uint32_t u;
    memcpy(&u, &f, 4);
    return u;
}

int main(int argc, char** argv)
{
    if (argc < 3) return 2;
    const std::string        out = argv[1];
    bool                     opencv = false;
    std::set<int>            dump;
    std::vector<std::string> files;
    for (int i = 2; i < argc; i++) {
        const std::string a = argv[i];
        if (a == "--opencv")
            opencv = true;
        else if (a == "--dump" && i + 1 < argc) {
            std::stringstream ss(argv[++i]);
            std::string       t;
            while (std::getline(ss, t, ',')) dump.insert(atoi(t.c_str()));
        } else
            files.push_back(a);
    }
    popsift::Config config;
    if (opencv) { /* BASELINE.json config 5: OpenCV-mode parameters */
        config.setMode(popsift::Config::OpenCV);
        config.setGaussMode("opencv");
    }
    struct Img {
        int            w, h;
        unsigned char* p;
    };
    std::vector<Img> imgs;
    for (const std::string& f : files) {
        Img im;
        im.p = readPGMfile(f, im.w, im.h);
        if (!im.p) return 3;
        imgs.push_back(im);
    }
    PopSift               sift(config);
    std::vector<SiftJob*> jobs;
    for (const Img& im : imgs) jobs.push_back(sift.enqueue(im.w, im.h, im.p)); /* all in flight at once */
    for (size_t i = 0; i < jobs.size(); i++) {
        popsift::Features*   f = jobs[i]->get();
        popsift::Descriptor* base = f->getDescriptors();
        uint64_t             digest = 0;
        for (auto it = f->begin(); it != f->end(); ++it) {
            const popsift::Feature& ft = *it;
            for (int k = 0; k < ft.num_ori; k++) {
                uint64_t h = 14695981039346656037ull;
                h = fnv(h, bits(ft.xpos));
                h = fnv(h, bits(ft.ypos));
                h = fnv(h, bits(ft.sigma));
                h = fnv(h, (uint32_t)ft.num_ori);
                h = fnv(h, (uint32_t)k);
                h = fnv(h, bits(ft.orientation[k]));
                for (int q = 0; q < 128; q++) h = fnv(h, bits(ft.desc[k]->features[q]));
                digest += h;
            }
        }
        printf("%zu %d %d %016llx\n", i, f->getFeatureCount(), f->getDescriptorCount(), (unsigned long long)digest);
        if (dump.count((int)i)) {
            const std::string path = out + "/result_" + std::to_string(i) + ".bin";
            FILE*             fp = fopen(path.c_str(), "wb");
            if (!fp) return 4;
            const int32_t nf = f->getFeatureCount(), nd = f->getDescriptorCount();
            fwrite(&nf, 4, 1, fp);
            fwrite(&nd, 4, 1, fp);
            for (auto it = f->begin(); it != f->end(); ++it) {
                const popsift::Feature& ft = *it;
                int32_t                 rec[4 + 4];
                float                   v[3 + 4] = {ft.xpos, ft.ypos, ft.sigma, ft.orientation[0], ft.orientation[1], ft.orientation[2],
                                                    ft.orientation[3]};
                rec[0] = ft.debug_octave;
                rec[1] = ft.num_ori;
                rec[2] = rec[3] = 0;
                for (int k = 0; k < 4; k++) rec[4 + k] = (k < ft.num_ori && ft.desc[k]) ? (int32_t)(ft.desc[k] - base) : -1;
                fwrite(v, 4, 7, fp);
                fwrite(rec, 4, 8, fp);
            }
            fwrite(base, 512, (size_t)nd, fp);
            fclose(fp);
        }
        delete f;
        delete jobs[i];
    }
    sift.uninit();
    for (Img& im : imgs) delete[] im.p;
    fflush(stdout);
    return 0;
}
