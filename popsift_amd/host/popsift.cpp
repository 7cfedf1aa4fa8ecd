/*
 * PopSift / SiftJob over the HIP C ABI (replaces popsift.cpp:16-318).
 *
 * The reference runs ONE pipeline: an upload thread, a compute thread and a single
 * process-global pyramid.  Here a PopSift object owns a pool of workers -- one per
 * extraction context, POPSIFT_CONTEXTS_PER_DEVICE contexts on each GPU listed in
 * POPSIFT_DEVICES -- that pull jobs from one queue (work stealing balances images
 * of different sizes and GPUs of different load).  Each worker drives its own
 * popsift_hip_ctx: submit (H2D + every kernel, asynchronous), wait, fetch into the
 * caller-visible FeaturesHost, fulfil the job's promise.  No collective is needed:
 * images are independent.
 */
#include "popsift/popsift.h"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <map>
#include <sstream>
#include <vector>

#include <pthread.h>
#include <sched.h>

#include "debug_dump.h"
#include "popsift/common/device_prop.h"
#include "popsift_hip.h"
#include "../csrc/trace.h" /* POPSIFT_RANGE: roctx ranges in a ROCTX=1 build (the reference: NVTX, popsift.h:20-25) */

using namespace std;

namespace {

[[noreturn]] void die(const char* file, int line, const std::string& msg)
{
    /* the reference's error convention: message on cerr, then exit (debug_macros.h:141-146) */
    cerr << file << ":" << line << endl << "E    " << msg << endl;
    exit(-1);
}
#define DIE(msg) die(__FILE__, __LINE__, (msg))

popsift_hip_params to_params(const popsift::Config& c)
{
    popsift_hip_params p;
    popsift_hip_default_params(&p);
    p.octaves = c.octaves;
    p.levels = c.levels;
    p.sigma = c.sigma;
    p.edge_limit = c._edge_limit;
    p.threshold = c.getThreshold();
    p.upscale_factor = c.getUpscaleFactor();
    p.sift_mode = (int)c.getSiftMode();
    p.gauss_mode = (int)c.getGaussMode();
    p.desc_mode = (int)c.getDescMode();
    p.norm_mode = (int)c.getNormMode();
    p.norm_multi = c.getNormalizationMultiplier();
    p.max_extrema = c.getMaxExtrema();
    p.assume_initial_blur = c.hasInitialBlur() ? 1 : 0;
    p.initial_blur = c.getInitialBlur();
    p.filter_grid_size = c.getFilterGridSize();
    p.filter_max_extrema = c.getFilterMaxExtrema();
    p.filter_sorting = (int)c.getFilterSorting(); /* RandomScale, LargestScaleFirst, SmallestScaleFirst */
    return p;
}

std::vector<int> device_list()
{
    int n = 0;
    if (popsift_hip_device_count(&n) != POPSIFT_HIP_OK || n <= 0) DIE("no usable GPU found");
    std::vector<int> devs;
    const char*      e = getenv("POPSIFT_DEVICES");
    if (!e || !*e) {
        /* device_prop_t::set(n) plays cudaSetDevice(n) (main.cpp:300-302): that GPU only */
        const int chosen = popsift::cuda::device_prop_t::chosenDevice();
        if (chosen >= 0 && chosen < n) {
            devs.push_back(chosen);
            return devs;
        }
    }
    if (!e || !*e || string(e) == "all") {
        for (int i = 0; i < n; i++) devs.push_back(i);
        return devs;
    }
    stringstream ss(e);
    string       tok;
    while (getline(ss, tok, ',')) {
        const int d = atoi(tok.c_str());
        if (d < 0 || d >= n) DIE("POPSIFT_DEVICES names a device that does not exist: " + tok);
        devs.push_back(d);
    }
    if (devs.empty()) DIE("POPSIFT_DEVICES is empty");
    return devs;
}

/* Config::setPrintGaussTables (--print-gauss-tables): the preamble of init_filter (gauss_filter.cu:147-163) and the
 * "relative sigma" block of print_gauss_filter_symbol (gauss_filter.cu:24-47), the one table this build applies
 * (the reference also prints its hardware-interpolation and absolute-filter tables, which do not exist here) */
void print_gauss_tables(const popsift::Config& conf, popsift_hip_ctx* ctx)
{
    int n = 0;
    if (popsift_hip_get_gauss_table(ctx, 0, 0, 0, &n) != POPSIFT_HIP_OK || n <= 0) return;
    std::vector<float> filter((size_t)n * POPSIFT_HIP_GAUSS_ALIGN), sigma((size_t)n);
    std::vector<int>   span((size_t)n);
    if (popsift_hip_get_gauss_table(ctx, filter.data(), span.data(), sigma.data(), &n) != POPSIFT_HIP_OK) return;
    printf("\n"
           "Upscaling factor: %f (i.e. original image is scaled by a factor of %f)\n"
           "\n"
           "Sigma computations\n"
           "    Initial sigma is %f\n"
           "    Input blurriness is assumed to be %f (scaled to %f)\n",
           conf.getUpscaleFactor(), pow(2.0f, conf.getUpscaleFactor()), conf.sigma, conf.getInitialBlur(),
           conf.getInitialBlur() * pow(2.0f, conf.getUpscaleFactor()));
    printf("\n"
           "Gauss tables\n"
           "      level span sigma : center value -> edge value\n"
           "    relative sigma\n");
    const int columns = 10;
    for (int lvl = 0; lvl < n; lvl++) {
        printf("      %d %d ", lvl, span[(size_t)lvl] + span[(size_t)lvl] - 1);
        printf("%2.6f: ", sigma[(size_t)lvl]);
        const int m = std::min(span[(size_t)lvl], columns);
        for (int x = 0; x < m; x++) printf("%0.8f ", filter[(size_t)lvl * POPSIFT_HIP_GAUSS_ALIGN + x]);
        printf(m < span[(size_t)lvl] ? "...\n" : "\n");
    }
    printf("\n");
    fflush(stdout);
}

std::atomic<int> g_live_pipelines{0}; /* PopSift objects with running workers */

}  // namespace

namespace popsift {
/* "0-15,128-143" (a sysfs cpulist) -> CPU set; returns the number of CPUs named.  Re-entrant: several workers parse at
 * once (strtok, which an earlier version used, keeps one process-wide state).  Exported for the unit test. */
int parseCpuList(const char* list, cpu_set_t* set)
{
    CPU_ZERO(set);
    int         n = 0;
    const char* p = list;
    while (p && *p) {
        while (*p == ',' || *p == ' ' || *p == '\n' || *p == '\t') p++;
        if (!*p) break;
        char*      end = 0;
        const long a = strtol(p, &end, 10);
        if (end == p) break; /* not a number: stop at what has been understood so far */
        long b = a;
        p = end;
        if (*p == '-') {
            b = strtol(p + 1, &end, 10);
            if (end == p + 1) break;
            p = end;
        }
        for (long c = std::max(a, 0L); c <= b && c < CPU_SETSIZE; c++)
            if (!CPU_ISSET((int)c, set)) {
                CPU_SET((int)c, set);
                n++;
            }
        while (*p && *p != ',') p++; /* e.g. a stride suffix ":2" is ignored */
    }
    return n;
}
}  // namespace popsift

namespace {

/* The CPUs of the NUMA node a GPU hangs off (one host thread per context, NUMA-local pinned buffers),
 * INTERSECTED with the affinity mask the process was started with: a launcher's taskset / numactl / MPI binding is
 * respected, and when it leaves no CPU of that node the worker is not bound at all.  POPSIFT_NUMA_BIND=0 switches the
 * binding off (the reference never touches affinity).  Returns false when there is nothing to apply. */
bool numa_cpu_set(int node, cpu_set_t* out)
{
    const char* e = getenv("POPSIFT_NUMA_BIND");
    if (e && atoi(e) == 0) return false;
    if (node < 0) return false;
    char path[96];
    snprintf(path, sizeof(path), "/sys/devices/system/node/node%d/cpulist", node);
    FILE* f = fopen(path, "r");
    if (!f) return false;
    char       list[4096] = {0};
    const bool ok = fgets(list, sizeof(list), f) != 0;
    fclose(f);
    if (!ok) return false;
    cpu_set_t node_set, cur;
    if (popsift::parseCpuList(list, &node_set) <= 0) return false;
    CPU_ZERO(&cur);
    if (pthread_getaffinity_np(pthread_self(), sizeof(cur), &cur) != 0) return false;
    CPU_AND(out, &node_set, &cur);
    return CPU_COUNT(out) > 0;
}

/* jobs of one size a worker takes from the queue at once (popsift_hip_submit_batch); it never waits for a batch to fill */
int jobs_per_submit()
{
    const char* e = getenv("POPSIFT_BATCH");
    const int   k = e ? atoi(e) : 1;
    return std::min(std::max(k, 1), POPSIFT_HIP_MAX_BATCH);
}

int contexts_per_device()
{
    const char* e = getenv("POPSIFT_CONTEXTS_PER_DEVICE");
    const int   k = e ? atoi(e) : 4; /* measured host to host with the in-context download overlap: 1 / 2 / 3 / 4 / 6 / 8 contexts = 1.60 / 1.90 / 2.06 / 2.03 / 2.02 / 1.82 Gpix/s (tools/h2h_sweep.sh) */
    return std::min(std::max(k, 1), 64);
}

}  // namespace

/* ------------------------------------------------------------------------- SiftJob */

/* The job's copy of the caller's image (the caller may free or reuse its buffer as soon as enqueue() returns,
 * popsift.cpp:245-247) goes into a recycled block of the pinned pool: no malloc -- a 2 MB malloc is an mmap and 512 page
 * faults every time -- and no second copy, the worker uploads straight from the block (popsift_hip_submit_pinned_*). */
SiftJob::SiftJob(int w, int h, const unsigned char* imageData) : _w(w), _h(h), _is_float(false)
{
    _f = _p.get_future();
    _imageData = (unsigned char*)popsift::pinnedBlockGet((size_t)w * h, &_pinned);
    if (_imageData == 0) DIE("Memory limitation: failed to allocate memory for SiftJob");
    memcpy(_imageData, imageData, (size_t)w * h);
}

SiftJob::SiftJob(int w, int h, const float* imageData) : _w(w), _h(h), _is_float(true)
{
    _f = _p.get_future();
    _imageData = (unsigned char*)popsift::pinnedBlockGet((size_t)w * h * sizeof(float), &_pinned);
    if (_imageData == 0) DIE("Memory limitation: failed to allocate memory for SiftJob");
    memcpy(_imageData, imageData, (size_t)w * h * sizeof(float));
}

SiftJob::~SiftJob() { popsift::pinnedBlockPut(_imageData); }

void SiftJob::setFeatures(popsift::FeaturesBase* f) { _p.set_value(f); }

popsift::FeaturesHost* SiftJob::get() { return getHost(); }
popsift::FeaturesBase* SiftJob::getBase() { return _f.get(); }
popsift::FeaturesHost* SiftJob::getHost() { return dynamic_cast<popsift::FeaturesHost*>(_f.get()); }
popsift::FeaturesDev*  SiftJob::getDev() { return dynamic_cast<popsift::FeaturesDev*>(_f.get()); }

/* ------------------------------------------------------------------------- PopSift */

PopSift::PopSift(const popsift::Config& config, popsift::Config::ProcessingMode mode, ImageMode imode)
    : _image_mode(imode)
{
    _proc_mode = mode;
    configure(config, true);
}

PopSift::PopSift(ImageMode imode) : _image_mode(imode) {}

PopSift::~PopSift()
{
    /* the reference leaks its threads if uninit() was forgotten (popsift.cpp:59-61); be kind */
    if (_started && !_stopped) uninit();
}

bool PopSift::configure(const popsift::Config& config, bool /*force*/)
{
    std::lock_guard<std::mutex> lk(_mtx);
    if (_started) return false; /* popsift.cpp:65-67: not after the pyramid exists */
    _config = config;
    _config.levels = max(2, config.levels); /* popsift.cpp:71 */
    /* validate now, fatally, like init_filter does (gauss_filter.cu:131-144) */
    if (_config.sigma > 2.0f) DIE("Sigma > 2.0 is not supported.");
    if (_config.levels > GAUSS_LEVELS - 3) DIE("More than 9 levels are not supported.");
    if (_config.getGaussMode() != popsift::Config::VLFeat_Compute &&
        _config.getGaussMode() != popsift::Config::OpenCV_Compute)
        DIE("this build implements the Gauss modes 'vlfeat' and 'opencv' only");
    if (_config.getScalingMode() != popsift::Config::ScaleDefault) DIE("ScaleDirect is not supported");
    if (_config.getFilterMaxExtrema() > 0 && (_config.getFilterGridSize() < 1 || _config.getFilterGridSize() > 64))
        DIE("the grid filter supports grid sizes 1..64");
    _shadow_config = _config;
    return true;
}

/* popsift.cpp:89-120: the first image fixes the octave count for every context */
void PopSift::start_workers(int w, int h)
{
    if (_config.octaves < 0) {
        const float scaleFactor = 1.0f / powf(2.0f, -_config.getUpscaleFactor());
        _config.octaves = max(int(floorf(logf((float)min(w, h)) / logf(2.0f)) - 3.0f + scaleFactor), 1);
    }
    const popsift_hip_params p = to_params(_config);
    const std::vector<int>   devs = device_list();
    const int                per = contexts_per_device();
    /* the CPU set of every device's NUMA node, worked out once, here, under the lock */
    std::map<int, std::pair<bool, cpu_set_t>> node_cpus;
    std::map<int, int>                        node_of;
    for (int d : devs) {
        if (node_of.count(d)) continue;
        int node = -1;
        if (popsift_hip_device_numa_node(d, &node) != POPSIFT_HIP_OK) node = -1;
        node_of[d] = node;
        cpu_set_t set;
        CPU_ZERO(&set);
        const bool have = numa_cpu_set(node, &set);
        node_cpus[d] = std::make_pair(have, set);
    }
    for (int k = 0; k < per; k++) {
        for (int d : devs) {
            Worker* wk = new Worker;
            wk->device = d;
            wk->numa_node = node_of[d];
            wk->bind = node_cpus[d].first;
            wk->cpus = node_cpus[d].second;
            const int rc = popsift_hip_ctx_create(d, &p, &wk->ctx);
            if (rc != POPSIFT_HIP_OK) DIE(string("cannot create extraction context: ") + popsift_hip_strerror(rc));
            _workers.push_back(wk);
        }
    }
    if (_config.ifPrintGaussTables() && !_workers.empty()) print_gauss_tables(_config, _workers[0]->ctx);
    for (Worker* wk : _workers) wk->thread = std::thread(&PopSift::worker_loop, this, wk);
    _started = true;
    g_live_pipelines++;
}

/* POD features -> popsift::Feature with descriptor pointers into the job's own block (prep_features, sift_pyramid.cu:249-279) */
static void convert_features(const popsift_hip_feature* pod, int nf, popsift::FeaturesHost* features)
{
    popsift::Feature*    out = features->getFeatures();
    popsift::Descriptor* base = features->getDescriptors();
    for (int i = 0; i < nf; i++) {
        const popsift_hip_feature& s = pod[i];
        popsift::Feature&          f = out[i];
        f.debug_octave = s.debug_octave;
        f.xpos = s.xpos;
        f.ypos = s.ypos;
        f.sigma = s.sigma;
        f.num_ori = s.num_ori;
        for (int k = 0; k < ORIENTATION_MAX_COUNT; k++) {
            f.orientation[k] = s.orientation[k];
            f.desc[k] = s.desc_idx[k] >= 0 ? base + s.desc_idx[k] : 0;
        }
    }
}

/* One worker = one context.  Where the reference's extractDownloadLoop (popsift.cpp:187-213) downloads image i before
 * it looks at image i+1, this loop starts the download (popsift_hip_fetch_begin: copy stream, second result slab) and
 * submits image i+1 first -- its kernels run under the PCIe transfer and under the host-side conversion of image i.
 * With nothing queued the pending download is completed at once, so a lone job sees no added latency. */
void PopSift::worker_loop(Worker* me)
{
    /* the image upload and every pinned block this worker allocates are then first touched on its GPU's node */
    if (me->bind) (void)pthread_setaffinity_np(pthread_self(), sizeof(me->cpus), &me->cpus);
    popsift::setPinnedPoolNode(me->numa_node); /* result blocks this thread takes come from / go to its node's free list */
    /* --log dumps read the context's planes after the image: keep those runs strictly serial */
    const bool overlap = _config.getLogMode() != popsift::Config::All;
    /* the jobs whose downloads are under way (popsift_hip_fetch_begin_item): completed after the next submit */
    struct Pending {
        SiftJob*               job = 0;
        popsift::FeaturesHost* features = 0;
        int                    nf = 0;
        int                    pod = 0; /* which staging buffer holds the POD features */
    };
    std::vector<Pending> pending;
    auto complete_pending = [&]() {
        if (pending.empty()) return;
        if (popsift_hip_fetch_end(me->ctx) != POPSIFT_HIP_OK) DIE(string("download failed: ") + popsift_hip_last_error(me->ctx));
        for (Pending& p : pending) {
            convert_features((const popsift_hip_feature*)me->pods[(size_t)p.pod].p, p.nf, p.features);
            p.job->setFeatures(p.features);
        }
        pending.clear();
    };
    /* pinned staging buffer number k for at least nf POD features (a pageable target would be staged by the runtime at a
     * fraction of the PCIe rate); the descriptors go directly into the caller-visible pinned block */
    auto pod_buffer = [&](size_t k, int nf) -> popsift_hip_feature* {
        if (me->pods.size() <= k) me->pods.resize(k + 1);
        Worker::Pod& b = me->pods[k];
        if ((size_t)nf > b.cap) {
            popsift_hip_host_free(b.p);
            b.cap = (size_t)nf + (size_t)nf / 4 + 1024;
            b.p = popsift_hip_host_alloc(b.cap * sizeof(popsift_hip_feature));
            if (!b.p) DIE("Memory limitation: failed to allocate the feature staging buffer");
        }
        return (popsift_hip_feature*)b.p;
    };
    const int max_batch = jobs_per_submit();
    for (;;) {
        SiftJob* job = 0;
        bool     quit = false;
        std::vector<SiftJob*> more; /* further queued jobs of the same size and type, extracted in the same submit */
        {
            std::unique_lock<std::mutex> lk(_mtx);
            if (pending.empty()) _cv.wait(lk, [&] { return !_queue.empty(); });
            if (!_queue.empty()) {
                job = _queue.front();
                if (job == 0)
                    quit = true; /* shutdown marker stays for the other workers */
                else {
                    _queue.pop();
                    while ((int)more.size() + 1 < max_batch && !_queue.empty() && _queue.front() != 0 &&
                           _queue.front()->getWidth() == job->getWidth() && _queue.front()->getHeight() == job->getHeight() &&
                           _queue.front()->isFloat() == job->isFloat() && _proc_mode != popsift::Config::MatchingMode &&
                           _config.getLogMode() != popsift::Config::All) {
                        more.push_back(_queue.front());
                        _queue.pop();
                    }
                }
            }
        }
        if (quit || !job) {
            complete_pending();
            if (quit) return;
            continue;
        }
        if (!more.empty()) {
            /* several jobs in one submit: every kernel is launched once for all of them (popsift_hip.h); their downloads
             * are started together and run under the next submit, like a single job's */
            POPSIFT_RANGE("PopSift jobs (submit_batch, wait, fetch)");
            std::vector<SiftJob*> jobs;
            jobs.push_back(job);
            jobs.insert(jobs.end(), more.begin(), more.end());
            bool pinned = true;
            for (SiftJob* j : jobs) pinned = pinned && j->isPinned();
            const void* imgs[POPSIFT_HIP_MAX_BATCH];
            for (size_t k = 0; k < jobs.size(); k++) imgs[k] = jobs[k]->getImageData();
            const int kind = job->isFloat() ? (pinned ? POPSIFT_HIP_IMG_PINNED_F32 : POPSIFT_HIP_IMG_HOST_F32)
                                            : (pinned ? POPSIFT_HIP_IMG_PINNED_U8 : POPSIFT_HIP_IMG_HOST_U8);
            int rc = popsift_hip_submit_batch(me->ctx, imgs, (int)jobs.size(), kind, job->getWidth(), job->getHeight(), job->getWidth());
            if (rc != POPSIFT_HIP_OK) DIE(string("extraction failed: ") + popsift_hip_last_error(me->ctx));
            complete_pending(); /* the previous jobs' downloads and conversion, under this batch's kernels */
            int n = 0, nfs[POPSIFT_HIP_MAX_BATCH], nds[POPSIFT_HIP_MAX_BATCH];
            rc = popsift_hip_wait_batch(me->ctx, &n, nfs, nds);
            if (rc != POPSIFT_HIP_OK || n != (int)jobs.size()) DIE(string("extraction failed: ") + popsift_hip_last_error(me->ctx));
            for (int k = 0; k < n; k++) {
                popsift::FeaturesHost* features = new popsift::FeaturesHost(nfs[k], nds[k]);
                if (nds[k] == 0) cerr << "Warning: no descriptors extracted" << endl; /* sift_desc.cu:88-92 */
                if (nfs[k] == 0) {
                    jobs[(size_t)k]->setFeatures(features);
                    continue;
                }
                popsift_hip_feature* pod = pod_buffer((size_t)k, nfs[k]);
                rc = popsift_hip_fetch_begin_item(me->ctx, k, pod, me->pods[(size_t)k].cap, (float*)features->getDescriptors(),
                                                  (size_t)nds[k] * 128);
                if (rc != POPSIFT_HIP_OK) DIE(string("download failed: ") + popsift_hip_last_error(me->ctx));
                Pending p;
                p.job = jobs[(size_t)k];
                p.features = features;
                p.nf = nfs[k];
                p.pod = k;
                pending.push_back(p);
            }
            continue;
        }
        POPSIFT_RANGE("PopSift job (submit, wait, fetch)");
        int rc;
        /* the job's block stays untouched until the job is deleted, i.e. beyond popsift_hip_wait below */
        if (job->isFloat())
            rc = (job->isPinned() ? popsift_hip_submit_pinned_f32 : popsift_hip_submit_f32)(
                me->ctx, (const float*)job->getImageData(), job->getWidth(), job->getHeight(), job->getWidth());
        else
            rc = (job->isPinned() ? popsift_hip_submit_pinned_u8 : popsift_hip_submit_u8)(
                me->ctx, job->getImageData(), job->getWidth(), job->getHeight(), job->getWidth());
        if (rc != POPSIFT_HIP_OK) DIE(string("extraction failed: ") + popsift_hip_last_error(me->ctx));
        complete_pending(); /* the previous image's download and conversion, under this image's kernels */
        int nf = 0, nd = 0;
        rc = popsift_hip_wait(me->ctx, &nf, &nd);
        if (rc != POPSIFT_HIP_OK) DIE(string("extraction failed: ") + popsift_hip_last_error(me->ctx));

        if (_proc_mode == popsift::Config::MatchingMode) {
            /* matchPrepareLoop (popsift.cpp:215-236): the results stay on the GPU */
            popsift_hip_devfeatures* set = 0;
            rc = popsift_hip_clone_results(me->ctx, &set);
            if (rc != POPSIFT_HIP_OK) DIE(string("cloning device results failed: ") + popsift_hip_last_error(me->ctx));
            job->setFeatures(new popsift::FeaturesDev(set));
            continue;
        }

        popsift::FeaturesHost* features = new popsift::FeaturesHost(nf, nd);
        if (nd == 0) cerr << "Warning: no descriptors extracted" << endl; /* sift_desc.cu:88-92 */
        if (nf > 0) {
            popsift_hip_feature* pod = pod_buffer(0, nf);
            if (overlap) {
                rc = popsift_hip_fetch_begin(me->ctx, pod, me->pods[0].cap, (float*)features->getDescriptors(), (size_t)nd * 128);
                if (rc != POPSIFT_HIP_OK) DIE(string("download failed: ") + popsift_hip_last_error(me->ctx));
                Pending p;
                p.job = job;
                p.features = features;
                p.nf = nf;
                p.pod = 0;
                pending.push_back(p);
                continue;
            }
            rc = popsift_hip_fetch(me->ctx, pod, me->pods[0].cap, (float*)features->getDescriptors(), (size_t)nd * 128);
            if (rc != POPSIFT_HIP_OK) DIE(string("download failed: ") + popsift_hip_last_error(me->ctx));
            convert_features(pod, nf, features);
        }
        if (_config.getLogMode() == popsift::Config::All) {
            /* popsift.cpp:201-209; with several workers the dumps of concurrent images overwrite
             * each other exactly as successive images do in the reference */
            static std::mutex           dump_mtx;
            std::lock_guard<std::mutex> lk(dump_mtx);
            popsift::debug::download_and_save_array(me->ctx, _config, "pyramid");
            popsift::debug::save_descriptors(_config, features, "pyramid");
        }
        job->setFeatures(features);
    }
}

void PopSift::uninit()
{
    {
        std::lock_guard<std::mutex> lk(_mtx);
        if (_stopped) return;
        _stopped = true;
        _queue.push(0);
    }
    _cv.notify_all();
    for (Worker* wk : _workers) {
        if (wk->thread.joinable()) wk->thread.join();
        popsift_hip_ctx_destroy(wk->ctx);
        for (Worker::Pod& b : wk->pods) popsift_hip_host_free(b.p);
        delete wk;
    }
    const bool had_workers = !_workers.empty();
    _workers.clear();
    /* the last pipeline of the process is gone: give the cached pinned result blocks back (features.cpp) */
    if (had_workers && --g_live_pipelines == 0) popsift::releasePinnedCache();
}

SiftJob* PopSift::enqueue(int w, int h, const unsigned char* imageData)
{
    if (_image_mode != ByteImages)
        DIE("Image mode error: cannot load byte images into a PopSift pipeline configured for float images");
    SiftJob* job = new SiftJob(w, h, imageData);
    {
        std::lock_guard<std::mutex> lk(_mtx);
        if (_stopped) DIE("enqueue() after uninit()");
        if (!_started) start_workers(w, h);
        _queue.push(job);
    }
    _cv.notify_one();
    return job;
}

SiftJob* PopSift::enqueue(int w, int h, const float* imageData)
{
    if (_image_mode != FloatImages)
        DIE("Image mode error: cannot load float images into a PopSift pipeline configured for byte images");
    SiftJob* job = new SiftJob(w, h, imageData);
    {
        std::lock_guard<std::mutex> lk(_mtx);
        if (_stopped) DIE("enqueue() after uninit()");
        if (!_started) start_workers(w, h);
        _queue.push(job);
    }
    _cv.notify_one();
    return job;
}
