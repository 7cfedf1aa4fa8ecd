// GPU test driver for BASELINE.json configs 4 and 5 through the drop-in C++ API (PopSift::enqueue ... SiftJob::get,
// the usage pattern of src/application/main.cpp:304-326): every PGM on the command line is enqueued into ONE PopSift
// object (its worker pool spans POPSIFT_DEVICES x POPSIFT_CONTEXTS_PER_DEVICE contexts), then every job is collected.
// Per image it prints "idx features descriptors digest" -- the digest is an order-independent 64-bit sum of FNV-1a
// hashes over (x, y, sigma, num_ori, k, orientation[k], the 128 descriptor words) of every orientation, which the
// Python side recomputes from a single-context C-ABI run of the same image -- and dumps the full result of the
// images named by --dump for the comparison with the oracle.
//   host_batch_test <out_dir> [--opencv] [--dump i,j,k] [--callers K] [--repeat R] [--window W] a.pgm b.pgm ...
// --callers K: K threads enqueue and collect (thread t takes jobs t, t + K, ...), each with at most W (default: all)
// of its jobs in flight; --repeat R: the file list R times over (job i is file i mod #files).
#include <popsift/features.h>
#include <popsift/popsift.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <cstring>
#include <set>
#include <sstream>
#include <string>
#include <thread>
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
    int                      callers = 1, repeat = 1, window = 0;
    std::vector<std::string> files;
    for (int i = 2; i < argc; i++) {
        const std::string a = argv[i];
        if (a == "--opencv")
            opencv = true;
        else if (a == "--dump" && i + 1 < argc) {
            std::stringstream ss(argv[++i]);
            std::string       t;
            while (std::getline(ss, t, ',')) dump.insert(atoi(t.c_str()));
        } else if (a == "--callers" && i + 1 < argc)
            callers = std::max(atoi(argv[++i]), 1);
        else if (a == "--repeat" && i + 1 < argc)
            repeat = std::max(atoi(argv[++i]), 1);
        else if (a == "--window" && i + 1 < argc)
            window = atoi(argv[++i]);
        else
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
    PopSift                  sift(config);
    const size_t             njobs = imgs.size() * (size_t)repeat;
    std::vector<std::string> lines(njobs);
    auto collect = [&](size_t i, SiftJob* job) {
        popsift::Features*   f = job->get();
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
        char buf[96];
        snprintf(buf, sizeof(buf), "%zu %d %d %016llx", i, f->getFeatureCount(), f->getDescriptorCount(), (unsigned long long)digest);
        lines[i] = buf;
        if (dump.count((int)i)) {
            const std::string path = out + "/result_" + std::to_string(i) + ".bin";
            FILE*             fp = fopen(path.c_str(), "wb");
            if (!fp) exit(4);
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
        delete job;
    };
    auto caller = [&](int t) {
        std::vector<std::pair<size_t, SiftJob*>> q; /* in flight, oldest first */
        size_t                                   head = 0;
        for (size_t i = (size_t)t; i < njobs; i += (size_t)callers) {
            const Img& im = imgs[i % imgs.size()];
            q.emplace_back(i, sift.enqueue(im.w, im.h, im.p));
            if (window > 0 && q.size() - head >= (size_t)window) {
                collect(q[head].first, q[head].second);
                head++;
            }
        }
        for (; head < q.size(); head++) collect(q[head].first, q[head].second);
    };
    std::vector<std::thread> th;
    for (int t = 1; t < callers; t++) th.emplace_back(caller, t);
    caller(0);
    for (auto& x : th) x.join();
    for (const std::string& l : lines) printf("%s\n", l.c_str());
    sift.uninit();
    for (Img& im : imgs) delete[] im.p;
    fflush(stdout);
    return 0;
}
