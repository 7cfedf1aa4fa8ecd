// GPU test of the drop-in C++ API under concurrency: several caller threads enqueue into one PopSift object
// (popsift.h: "enqueue is callable from any thread"), two PopSift objects live at once, results per job equal the
// single-threaded reference run, MatchingMode alongside ExtractingMode.
#include <popsift/features.h>
#include <popsift/popsift.h>

#include <cstdio>
#include <cstdlib>
#include <random>
#include <thread>
#include <vector>

#define CHECK(c)                                                                      \
    do {                                                                              \
        if (!(c)) {                                                                   \
            std::fprintf(stderr, "CHECK failed %s:%d: %s\n", __FILE__, __LINE__, #c); \
            std::exit(1);                                                             \
        }                                                                             \
    } while (0)

static std::vector<unsigned char> image(int w, int h, unsigned seed)
{
    std::mt19937               rng(seed);
    std::vector<unsigned char> v((size_t)w * h);
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            const int blob = ((x / 9 + y / 7 + (int)seed) % 5 == 0) ? 90 : 0;
            v[(size_t)y * w + x] = (unsigned char)(60 + blob + (int)(rng() % 60));
        }
    return v;
}

// host_mt_test [callers [jobs per caller]]   (default 4 x 12; the worker pool comes from POPSIFT_DEVICES /
// POPSIFT_CONTEXTS_PER_DEVICE, e.g. POPSIFT_DEVICES=0,0 for two workers' worth of contexts on one card)
int main(int argc, char** argv)
{
    const int                               W = 320, H = 240, NIMG = 6;
    const int                               CALLERS = argc > 1 ? std::atoi(argv[1]) : 4, PER = argc > 2 ? std::atoi(argv[2]) : 12;
    std::vector<std::vector<unsigned char>> imgs;
    for (int i = 0; i < NIMG; i++) imgs.push_back(image(W, H, 17u + (unsigned)i));

    popsift::Config config;
    // reference counts, one job at a time
    std::vector<int> nf(NIMG), nd(NIMG);
    {
        PopSift sift(config);
        for (int i = 0; i < NIMG; i++) {
            SiftJob*           j = sift.enqueue(W, H, imgs[(size_t)i].data());
            popsift::Features* f = j->get();
            nf[(size_t)i] = f->getFeatureCount();
            nd[(size_t)i] = f->getDescriptorCount();
            CHECK(nf[(size_t)i] > 50);
            delete f;
            delete j;
        }
        sift.uninit();
    }
    // CALLERS caller threads x PER jobs into one object, a second object running at the same time
    PopSift                  a(config), b(config, popsift::Config::MatchingMode);
    std::vector<std::thread> th;
    for (int t = 0; t < CALLERS; t++)
        th.emplace_back([&, t] {
            std::vector<SiftJob*> jobs;
            std::vector<int>      which;
            for (int k = 0; k < PER; k++) {
                const int i = (t * 5 + k) % NIMG;
                jobs.push_back(a.enqueue(W, H, imgs[(size_t)i].data()));
                which.push_back(i);
            }
            for (size_t k = 0; k < jobs.size(); k++) {
                popsift::Features* f = jobs[k]->get();
                CHECK(f->getFeatureCount() == nf[(size_t)which[k]] && f->getDescriptorCount() == nd[(size_t)which[k]]);
                int seen = 0;
                for (const popsift::Feature& ft : *f) seen += ft.num_ori;
                CHECK(seen == f->getDescriptorCount());
                delete f;
                delete jobs[k];
            }
        });
    SiftJob*              j0 = b.enqueue(W, H, imgs[0].data());
    SiftJob*              j1 = b.enqueue(W, H, imgs[0].data());
    popsift::FeaturesDev* d0 = j0->getDev();
    popsift::FeaturesDev* d1 = j1->getDev();
    CHECK(d0 && d1 && d0->getDescriptorCount() == nd[0] && d1->getFeatureCount() == nf[0]);
    const std::vector<popsift::FeaturesDev::Match> m = d0->matchAndGet(d1);
    CHECK((int)m.size() == nd[0]);
    int self = 0;
    for (const auto& x : m) self += (x.dist_best == 0.0f) ? 1 : 0; /* the same image twice: every descriptor finds itself */
    CHECK(self == nd[0]);
    for (std::thread& t : th) t.join();
    delete d0;
    delete d1;
    delete j0;
    delete j1;
    a.uninit();
    CHECK(popsift::pinnedCacheBytes() > 0); /* b still runs: the pool keeps its free blocks */
    b.uninit();
    CHECK(popsift::pinnedCacheBytes() == 0); /* the last pipeline is gone: no pinned memory stays behind */
    std::printf("host_mt_test ok\n");
    return 0;
}
